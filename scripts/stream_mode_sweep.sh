#!/bin/bash
set -e
timeout -k 10 400 python -c "
import cProfile, pstats, runpy, sys, io
sys.argv = ['bench_average.py', '128', '32', '8']
pr = cProfile.Profile()
pr.enable()
runpy.run_path('scripts/bench_average.py', run_name='__main__')
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(40)
print(s.getvalue())
" 2>&1 | tail -75 > gpurun_out/r2_average_profile.txt
tail -62 gpurun_out/r2_average_profile.txt | cut -c1-170
