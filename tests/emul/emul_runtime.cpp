// Runtime of the CPU emulation (tests/emul/hip/hip_runtime.h)  --  TEST INFRASTRUCTURE ONLY.
// Blocks of a launch are distributed over a few OS worker threads; inside a block the GPU threads are
// ucontext fibers scheduled round-robin, yielding at barriers and wave collectives.
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <ucontext.h>

namespace emul {
thread_local Worker* W = nullptr;
thread_local dim3 t_idx;
thread_local int t_linear = 0;
thread_local dim3 b_idx;
thread_local unsigned char* dyn_smem = nullptr;
dim3 b_dim, g_dim;

constexpr size_t STACK = 128 * 1024;
enum { READY = 0, WAIT_BLOCK = 1, WAIT_WAVE = 2, DONE = 3 };

struct Fiber {
    ucontext_t ctx;
    char* stack = nullptr;
    int state = READY;
    unsigned long long wait_gen = 0;
};

struct Worker {
    ucontext_t sched;
    std::vector<Fiber> fibers;
    int n = 0, current = 0;
    int arrived = 0, live = 0;
    unsigned long long gen = 0;
    std::vector<int> w_arrived, w_live;
    std::vector<unsigned long long> w_gen;
    std::vector<WaveBuf> w_buf;
    const std::function<void()>* body = nullptr;
    unsigned char* smem = nullptr;
};

static thread_local long long idle_yields = 0;

static void yield_to_scheduler() {
    Worker* w = W;
    swapcontext(&w->fibers[w->current].ctx, &w->sched);
}

void block_barrier() {
    Worker* w = W;
    Fiber& f = w->fibers[w->current];
    if (++w->arrived == w->live) {
        w->arrived = 0;
        ++w->gen;
        idle_yields = 0;
        return;
    }
    f.wait_gen = w->gen;
    f.state = WAIT_BLOCK;
    yield_to_scheduler();
}

void wave_barrier() {
    Worker* w = W;
    Fiber& f = w->fibers[w->current];
    const int wv = w->current / WAVE;
    if (++w->w_arrived[wv] == w->w_live[wv]) {
        w->w_arrived[wv] = 0;
        ++w->w_gen[wv];
        return;
    }
    f.wait_gen = w->w_gen[wv];
    f.state = WAIT_WAVE;
    yield_to_scheduler();
}

WaveBuf& wave_buf() { return W->w_buf[W->current / WAVE]; }

// spin waits: the fiber stays runnable; a block in which nothing but yields happens for a long time is livelocked
void fiber_yield() {
    if (++idle_yields > 200000000ll) {
        std::fprintf(stderr, "emul: LIVELOCK: threads spin in s_sleep waits and nothing else makes progress\n");
        std::abort();
    }
    yield_to_scheduler();
}

unsigned long long wave_ballot(int pred) {
    Worker* w = W;
    const int me = w->current, wv = me / WAVE;
    w->w_buf[wv].i[me % WAVE] = pred ? 1 : 0;
    wave_barrier();
    w = W;
    unsigned long long m = 0;
    for (int l = wv * WAVE; l < (wv + 1) * WAVE && l < w->n; ++l)
        if (w->fibers[l].state != DONE && w->w_buf[wv].i[l % WAVE]) m |= 1ull << (l % WAVE);
    wave_barrier();
    return m;
}

int wave_all(int pred) {
    Worker* w = W;
    const int me = w->current, wv = me / WAVE;
    w->w_buf[wv].i[me % WAVE] = pred ? 1 : 0;
    wave_barrier();
    w = W;
    int all = 1;
    for (int l = wv * WAVE; l < (wv + 1) * WAVE && l < w->n; ++l)
        if (w->fibers[l].state != DONE && !w->w_buf[wv].i[l % WAVE]) all = 0;
    wave_barrier();
    return all;
}

static void fiber_main() {
    Worker* w = W;
    (*w->body)();
    idle_yields = 0;
    w = W;
    Fiber& f = w->fibers[w->current];
    f.state = DONE;
    const int wv = w->current / WAVE;
    // a finished thread no longer takes part in barriers (GPU semantics of an early return)
    if (--w->live > 0 && w->arrived == w->live) {
        w->arrived = 0;
        ++w->gen;
    }
    if (--w->w_live[wv] > 0 && w->w_arrived[wv] == w->w_live[wv]) {
        w->w_arrived[wv] = 0;
        ++w->w_gen[wv];
    }
    swapcontext(&f.ctx, &w->sched);
}

static void run_block(Worker* w, int n, dim3 bidx, size_t smem_bytes) {
    if ((int)w->fibers.size() < n) {
        size_t old = w->fibers.size();
        w->fibers.resize(n);
        for (size_t i = old; i < (size_t)n; ++i) {
            void* p = mmap(nullptr, STACK, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
            if (p == MAP_FAILED) {
                std::fprintf(stderr, "emul: cannot allocate fiber stack\n");
                std::abort();
            }
            w->fibers[i].stack = (char*)p;
        }
    }
    const int n_waves = (n + WAVE - 1) / WAVE;
    w->n = n;
    w->live = n;
    w->arrived = 0;
    w->w_arrived.assign(n_waves, 0);
    w->w_live.assign(n_waves, 0);
    w->w_gen.assign(n_waves, 0);
    w->w_buf.resize(n_waves);
    for (int i = 0; i < n; ++i) w->w_live[i / WAVE]++;
    b_idx = bidx;
    dyn_smem = w->smem;
    std::memset(w->smem, 0xA5, smem_bytes);          // LDS is uninitialised on a GPU
    // the hardware drops LDS stores beyond the workgroup's allocation (and returns 0 for loads): a canary behind the requested
    // bytes catches such stores here, where they would otherwise land in the slack of the 160 KiB buffer unnoticed
    std::memset(w->smem + smem_bytes, 0x5C, DYN_SMEM_GUARD);
    for (int i = 0; i < n; ++i) {
        Fiber& f = w->fibers[i];
        getcontext(&f.ctx);
        f.ctx.uc_stack.ss_sp = f.stack;
        f.ctx.uc_stack.ss_size = STACK;
        f.ctx.uc_link = &w->sched;
        f.state = READY;
        makecontext(&f.ctx, (void (*)())fiber_main, 0);
    }
    int remaining = n;
    while (remaining > 0) {
        bool progress = false;
        for (int i = 0; i < n; ++i) {
            Fiber& f = w->fibers[i];
            if (f.state == DONE) continue;
            if (f.state == WAIT_BLOCK && f.wait_gen == w->gen) continue;
            if (f.state == WAIT_WAVE && f.wait_gen == w->w_gen[i / WAVE]) continue;
            f.state = READY;
            w->current = i;
            t_linear = i;
            t_idx.x = i % b_dim.x;
            t_idx.y = (i / b_dim.x) % b_dim.y;
            t_idx.z = i / (b_dim.x * b_dim.y);
            swapcontext(&w->sched, &f.ctx);
            progress = true;
            if (f.state == DONE) --remaining;
        }
        if (!progress && remaining > 0) {
            std::fprintf(stderr, "emul: DEADLOCK in block (%u,%u,%u): %d threads wait at a barrier that the others "
                                 "never reach (divergent __syncthreads / wave collective)\n", bidx.x, bidx.y, bidx.z, remaining);
            std::abort();
        }
    }
    for (size_t i = smem_bytes; i < smem_bytes + DYN_SMEM_GUARD; ++i)
        if (w->smem[i] != 0x5C) {
            std::fprintf(stderr, "emul: block (%u,%u,%u) stored to dynamic LDS at byte %zu, beyond its allocation of %zu bytes\n",
                         bidx.x, bidx.y, bidx.z, i, smem_bytes);
            std::abort();
        }
}

static std::mutex launch_mutex;

void launch(dim3 grid, dim3 block, size_t smem, const std::function<void()>& body) {
    std::lock_guard<std::mutex> guard(launch_mutex);
    if (smem > DYN_SMEM_MAX) {
        std::fprintf(stderr, "emul: dynamic LDS request %zu exceeds 160 KiB\n", smem);
        std::abort();
    }
    const int n = (int)(block.x * block.y * block.z);
    if (n <= 0 || n > 1024) {
        std::fprintf(stderr, "emul: bad block size %d\n", n);
        std::abort();
    }
    b_dim = block;
    g_dim = grid;
    const long long nblocks = (long long)grid.x * grid.y * grid.z;
    static int n_workers = [] {
        const char* e = std::getenv("MTIP_EMUL_THREADS");
        int v = e ? std::atoi(e) : (int)std::thread::hardware_concurrency();
        return v < 1 ? 1 : (v > 16 ? 16 : v);
    }();
    const int nw = (int)std::min<long long>(n_workers, nblocks);
    std::atomic<long long> next{0};
    static std::vector<Worker*> pool;                 // persistent: fiber stacks are reused across launches
    while ((int)pool.size() < nw) {
        Worker* w = new Worker();
        w->smem = (unsigned char*)aligned_alloc(64, DYN_SMEM_MAX + DYN_SMEM_GUARD);
        pool.push_back(w);
    }
    auto work = [&](int id) {
        Worker* mine = pool[id];
        W = mine;
        mine->body = &body;
        for (;;) {
            const long long b = next.fetch_add(1);
            if (b >= nblocks) break;
            dim3 bidx((unsigned)(b % grid.x), (unsigned)((b / grid.x) % grid.y), (unsigned)(b / ((long long)grid.x * grid.y)));
            run_block(mine, n, bidx, smem);
        }
    };
    if (nw <= 1) {
        work(0);
    } else {
        std::vector<std::thread> ts;
        for (int i = 0; i < nw; ++i) ts.emplace_back(work, i);
        for (auto& t : ts) t.join();
    }
}
}  // namespace emul

// tells the Python side that "device" pointers of this library are host pointers (torch CPU tensors, not cuda ones)
extern "C" int mtip_emulated(void) { return 1; }
