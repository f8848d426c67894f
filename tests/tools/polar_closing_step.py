"""numpy prototype of the closing step of k_rproj: one-sided Jacobi sweeps on realistic real-form
Procrustes matrices (warm start from the previous intensity), then the polar factor from the Gram matrix to first and to second
order in the remaining non-orthogonality E -- error against the SVD route per sweep.  Imports the oracle: a study script, not product."""
import sys, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
from helpers import OracleTransforms, golden_settings
from oracle.fourier import FourierPair
from oracle.sht import SHT
from oracle import mtip as OM
from xframe_amd.fxs import synthetic as S
N,L=40,18
sht=SHT(L); fpd=FourierPair(sht,N,S.data_cutoff(N),2.0)
data,_=S.make_invariants(OracleTransforms(fpd),N,L)
opt=golden_settings(N,L); om=OM.MTIP(opt,data)
rng=np.random.default_rng(0)
def real_M(l, grid):
    Ilm=sht.forward_l(grid.astype(complex))
    I=Ilm[l]; n=2*l+1
    V=om.rp.projection_matrices[l].real
    q=om.rp.radial_points
    # real form: columns sqrt2 Re, sqrt2 Im of m>=0
    cols=[I[:,l].real]
    for m in range(1,l+1):
        cols += [np.sqrt(2)*I[:,l+m].real, np.sqrt(2)*I[:,l+m].imag]
    It=np.stack(cols,1)
    return V.T@(q[:,None]**2*It)      # k x n
def sweep(W,Vr):
    k=W.shape[1]; rmax=0
    for i in range(k-1):
        for j in range(i+1,k):
            a=W[:,i]@W[:,i]; b=W[:,j]@W[:,j]; g=W[:,i]@W[:,j]
            if g*g <= 1e-30*a*b or g==0: continue
            rmax=max(rmax, abs(g)/np.sqrt(a*b))
            d=0.5*(b-a); t=np.sign(d if d!=0 else 1)*g/(abs(d)+np.sqrt(d*d+g*g)); c=1/np.sqrt(1+t*t); s=c*t
            wi=c*W[:,i]-s*W[:,j]; wj=s*W[:,i]+c*W[:,j]; W[:,i],W[:,j]=wi,wj
            vi=c*Vr[:,i]-s*Vr[:,j]; vj=s*Vr[:,i]+c*Vr[:,j]; Vr[:,i],Vr[:,j]=vi,vj
    return rmax
def polar_exact(M):
    u,s,vh=np.linalg.svd(M,full_matrices=False); return u@vh, s
def finish(W,Vr,tabs):
    sig=np.linalg.norm(W,axis=0); ok=sig>tabs
    Z=np.where(ok, W/np.where(ok,sig,1), 0)
    return Z@Vr.T    # (n x k)(k x k) -> U^T  (n x k)
def finish_corr(W,Vr,tabs):
    G=W.T@W; sig=np.sqrt(np.diag(G)); ok=sig>tabs
    isg=np.where(ok,1/np.where(ok,sig,1),0)
    S=sig[:,None]+sig[None,:]
    C=np.where(ok[:,None]&ok[None,:], G/np.where(S>0,S,1)*isg[None,:], 0); np.fill_diagonal(C,0)   # (Delta D^-1)_ij = G_ij/((si+sj) sj)
    E=np.abs(G*isg[:,None]*isg[None,:]-np.diag(ok.astype(float))).max()
    Z=(W*isg[None,:])@(np.eye(len(sig))-C)
    return Z@Vr.T, E
g0=rng.random((N,sht.n_theta,sht.n_phi))
g1=g0+0.02*rng.random(g0.shape)        # "next step": a nearby intensity
for l in (6,12,18):
    M0=real_M(l,g0).T    # n x k  (rows rho', columns j): one-sided Jacobi on columns of X~ = M^T
    M1=real_M(l,g1).T
    k=M0.shape[1]
    # converge on M0 to get warm V
    W=M0.copy(); Vr=np.eye(k)
    for s in range(12):
        if sweep(W,Vr)<1e-14: break
    Uex,sv=polar_exact(M1.T)   # polar of M (k x n): U = u vh (k x n)
    print(f'l={l} k={k} sigma range {sv.max():.2e} .. {sv.min():.2e}')
    W=M1@Vr; V2=Vr.copy()
    tabs=1e-13*sv.max()
    for s in range(6):
        r=sweep(W,V2)
        Ut=finish(W,V2,tabs); err=np.abs(Ut.T-Uex).max()
        Utc,E=finish_corr(W,V2,tabs); errc=np.abs(Utc.T-Uex).max()
        print(f'   sweep {s+1}: max rotation {r:.1e}  E after {E:.1e}  | err plain finish {err:.1e}  with first-order correction {errc:.1e}')

print('---- second order ----')
def finish_corr2(W,Vr,tabs):
    G=W.T@W; sig=np.sqrt(np.diag(G)); ok=sig>tabs
    isg=np.where(ok,1/np.where(ok,sig,1),0)
    S=sig[:,None]+sig[None,:]; Sinv=np.where(S>0,1/np.where(S>0,S,1),0)
    O=G.copy(); np.fill_diagonal(O,0); O=np.where(ok[:,None]&ok[None,:],O,0)
    A=O*isg[:,None]*isg[None,:]*Sinv
    Ab=A*sig[None,:]
    P=np.diag(isg)-A+(Ab@Ab.T)*Sinv+Ab@A
    E=np.abs(O*isg[:,None]*isg[None,:]).max()
    Z=W@P
    return Z@Vr.T, E
for l in (6,12):
    M0=real_M(l,g0).T; M1=real_M(l,g1).T; k=M0.shape[1]
    W=M0.copy(); Vr=np.eye(k)
    for s_ in range(12):
        if sweep(W,Vr)<1e-14: break
    Uex,sv=polar_exact(M1.T)
    W=M1@Vr; V2=Vr.copy(); tabs=1e-13*sv.max()
    for s_ in range(4):
        r=sweep(W,V2)
        U1,E=finish_corr(W,V2,tabs); U2,_=finish_corr2(W,V2,tabs)
        print(f'l={l} sweep {s_+1}: r {r:.1e} E after {E:.1e} | first order {np.abs(U1.T-Uex).max():.1e}  second order {np.abs(U2.T-Uex).max():.1e}')
