#!/bin/bash
# Round 6, evidence at HEAD for every record the default bench line prints: one tools/profile.sh run (kernel stats + PMC passes +
# calibration) per workload, the traffic stamped into profiles/pmc_traffic.json with the hash of the kernel source; the
# matrix-core counters of the two opt-in MFMA forms of the banded SpMM.  Usage: tools/exp_r06_evidence.sh [which ...]  (default: all)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
WHICH=${*:-"cfg2 plain poisson cfg4 cfg3 banded cfg5 csc add spgemm4 transpose sptrsv mfma"}
for w in $WHICH; do
  case $w in
    cfg2) bash tools/profile.sh r06x > gpurun_out/ev_$w.log 2>&1; python3 tools/stamp_traffic.py r06x >> gpurun_out/ev_$w.log 2>&1;;
    plain) bash tools/profile.sh r06p --workload spmv_plain > gpurun_out/ev_$w.log 2>&1
           python3 tools/stamp_traffic.py r06p spmv_plain_cfg2 spmv_sliced.hip 'pb_expand_kernel<float' 'pb_reduce_vf_kernel<float' >> gpurun_out/ev_$w.log 2>&1;;
    poisson) bash tools/profile.sh r06pp --workload spmv_poisson1 > gpurun_out/ev_$w.log 2>&1
           python3 tools/stamp_traffic.py r06pp spmv_poisson_cfg2 spmv_sliced.hip 'pb_expand_kernel<float' 'pb_reduce_kernel<float' >> gpurun_out/ev_$w.log 2>&1;;
    cfg4) bash tools/profile.sh r06x4 --workload spmv_rmat1 > gpurun_out/ev_$w.log 2>&1
          python3 tools/stamp_traffic.py r06x4 spmv_rmat spmv_sliced.hip 'pb_expand_kernel<double' 'pb_reduce_kernel<double' 'pb_split_finish_kernel<double' 'pb_empty_rows_kernel<double' 'pb_hot_rows_kernel<double' 'pb_hot_fixup_kernel<double' >> gpurun_out/ev_$w.log 2>&1;;
    cfg3) bash tools/profile.sh r06a3 --workload spmm > gpurun_out/ev_$w.log 2>&1
          python3 tools/stamp_traffic.py r06a3 spmm_cfg3 spmm.hip 'spmm_rowgroup_kernel<float' >> gpurun_out/ev_$w.log 2>&1;;
    banded) bash tools/profile.sh r06b3 --workload spmm_banded > gpurun_out/ev_$w.log 2>&1
          python3 tools/stamp_traffic.py r06b3 spmm_banded spmm.hip 'spmm_band_kernel' 'spmm_rowgroup_kernel<float' >> gpurun_out/ev_$w.log 2>&1;;
    cfg5) bash tools/profile.sh r06r5 --workload spgemm > gpurun_out/ev_$w.log 2>&1
          python3 tools/stamp_traffic.py r06r5 spgemm_cfg5 spgemm.hip 'spg_pack_b_kernel' 'spg_direct_kernel<float, true, false, false>' 'spg_direct_kernel<float, true, true, false>' >> gpurun_out/ev_$w.log 2>&1;;
    csc) bash tools/profile.sh r06cs --workload csc_spmv > gpurun_out/ev_$w.log 2>&1
          python3 tools/stamp_traffic.py r06cs csc_spmv_8f spmv_sliced.hip 'pb_expand_kernel<float' 'pb_reduce_kernel<float' >> gpurun_out/ev_$w.log 2>&1;;
    add) bash tools/profile.sh r06ad --workload add > gpurun_out/ev_$w.log 2>&1
          python3 tools/stamp_traffic.py r06ad add_8f spgemm.hip 'spg_ranked_fill_kernel' >> gpurun_out/ev_$w.log 2>&1;;
    spgemm4) bash tools/profile.sh r06g4 --workload spgemm4 > gpurun_out/ev_$w.log 2>&1
          python3 tools/stamp_traffic.py r06g4 spgemm4_8f spgemm.hip 'spg_pack_b_kernel' 'spg_direct_kernel<float, true, false, true>' 'spg_direct_kernel<float, true, true, true>' >> gpurun_out/ev_$w.log 2>&1;;
    transpose) bash tools/profile.sh r06tr --workload transpose > gpurun_out/ev_$w.log 2>&1
          python3 tools/stamp_traffic.py r06tr transpose_8f transpose.hip 'spt_tile_rows_kernel' 'spt_count_kernel*3' 'spt_scatter_kernel<float, 8, true>' 'spt_scatter_kernel<float, 8, false>*2' 'scan_block_sums_kernel*3' 'scan_partials_kernel*3' 'scan_apply_kernel*3' 'spt_rowptr_fill_kernel' 'spt_rowptr_long_kernel' >> gpurun_out/ev_$w.log 2>&1;;
    sptrsv) bash tools/profile.sh r06ts --workload sptrsv > gpurun_out/ev_$w.log 2>&1
          python3 tools/stamp_traffic.py r06ts sptrsv_8f sptrsv.hip 'trsv_level_kernel<float, 8>*158' 'trsv_chain_kernel<float, 8>*3' >> gpurun_out/ev_$w.log 2>&1;;
    mfma) SPBLAS_GFX950_SPMM_BAND=0 bash tools/prof_mfma.sh r06tiles > gpurun_out/ev_${w}_tiles.log 2>&1; tail -2 gpurun_out/ev_${w}_tiles.log
          SPBLAS_GFX950_SPMM_BAND_DENSE=250 bash tools/prof_mfma.sh r06window > gpurun_out/ev_$w.log 2>&1;;
  esac
  tail -3 gpurun_out/ev_$w.log
done
