#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python3 scripts/config5_steps.py 2 6 > gpurun_out/r3_cfg5_B2.txt 2>&1; cat gpurun_out/r3_cfg5_B2.txt
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r3_c5 -- python3 scripts/config5_steps.py 2 6 > /dev/null 2>&1
python scripts/kernel_stats.py gpurun_out/prof_r3_c5 > gpurun_out/r3_cfg5_B2_kernel_stats.txt; cat gpurun_out/r3_cfg5_B2_kernel_stats.txt
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F64 --output-format csv -d gpurun_out/pmc_r3_c5_mfma -- python3 scripts/config5_steps.py 2 6 > /dev/null 2>&1
python scripts/pmc_mfma.py gpurun_out/pmc_r3_c5_mfma > gpurun_out/r3_cfg5_pmc_mfma.txt 2>&1; cat gpurun_out/r3_cfg5_pmc_mfma.txt
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F64 --output-format csv -d gpurun_out/pmc_r3_c4_mfma -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > /dev/null 2>&1
python scripts/pmc_mfma.py gpurun_out/pmc_r3_c4_mfma > gpurun_out/r3_cfg4_pmc_mfma.txt 2>&1; cat gpurun_out/r3_cfg4_pmc_mfma.txt
