cd $GRAFT_REPO_ROOT
bash tools/pmc_bound.sh r04l > gpurun_out/r04l_pmc.log 2>&1
bash tools/pmc_cmd.sh r04l "python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-parity --path-only --repeats 1 --steps 40 --warmup 8" "integrate_kernel<false, false\|process_ray_kernel<false, false\|process_ray_tail\|bilateral_kernel" "l2:TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" "mem:TA_TA_BUSY_sum TA_BUSY_avr SQ_LDS_BANK_CONFLICT SQ_INST_LEVEL_VMEM" >> gpurun_out/r04l_pmc.log 2>&1
tail -40 gpurun_out/r04l_pmc.log
