#!/bin/bash
# round 6, visit w: LDS bank-conflict attribution of conv_x3r_kernel by instruction class (probe builds that drop, in turn, the producers'
# patch stores / the fragment reads / the K-quarter partial tiles / the transposing epilogue), counters per launch
O=$PWD/gpurun_out; mkdir -p $O; R=$PWD
cd /tmp && export TMPDIR=/tmp
for shape in "32 128 32" "32 192 64"; do
  tag=$(echo $shape | tr ' ' '_')
  args=""
  for v in base nostore noa nored noepi none; do
    rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS --kernel-trace --output-format csv -d /tmp/lds_${v}_$tag -- $R/tools/x3r_l_$v $shape > /tmp/lds_${v}_$tag.log 2>&1 || tail -5 /tmp/lds_${v}_$tag.log
    args="$args $v=/tmp/lds_${v}_$tag"
  done
  echo "== N Cin Cout = $shape"
  python $R/tools/lds_attr.py conv_x3r_kernel $args
done > $O/r06w_x3r_lds_attribution.txt 2>&1
cat $O/r06w_x3r_lds_attribution.txt
