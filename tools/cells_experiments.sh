#!/bin/bash
# The round-6 experiments on cast_cells_kernel, as they were run (each block one gpurun call; variants built here first with
# tools/ab_one_file.sh <name> raycast "<flags>" from a tree with the patch of tools/experiments/ applied).  Results: profiles/r06_cells_*.txt
#   knock-outs:     for k in 0..5: tools/ab_one_file.sh ko$k raycast -DTSDF_CELLS_KO=$k;  tools/ab_one_file.sh mix raycast -DTSDF_DIAG_RAY_MIX
case $1 in
knockouts)
  for g in 512 256; do for k in 0 1 2 3 4 5 0; do echo -n "KO$k "; TSDF_HIP_LIB=$GRAFT_REPO_ROOT/build/variants/ko$k/libtsdf_hip.so python tools/dbg_ray_cells.py 40 $g 2>&1 | tail -1; done; done
  TSDF_HIP_LIB=$GRAFT_REPO_ROOT/build/variants/mix/libtsdf_hip.so python tools/dbg_ray_mix.py 40 512 | tail -3 ;;
pmc)
  for k in 0 1 2 3 4 5; do TSDF_HIP_LIB=$GRAFT_REPO_ROOT/build/variants/ko$k/libtsdf_hip.so bash tools/pmc_cmd.sh ko$k "python $GRAFT_REPO_ROOT/tools/dbg_ray_cells.py 40 512" cast_cells "insts:SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVES"; done
  bash tools/pmc_cmd.sh lanes "python $GRAFT_REPO_ROOT/tools/dbg_ray_cells.py 40 512" cast_cells "lanes:SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" ;;
grid)
  for g in 512 256; do for n in 512 768 1024 1536 2048 2560 3072 3584 4096 8192; do echo -n "GRID=$n "; TSDF_RAY_CELLS_GRID=$n python tools/dbg_ray_cells.py 40 $g 2>&1 | tail -1; done; done ;;
*) echo "usage: $0 knockouts | pmc | grid" ;;
esac
