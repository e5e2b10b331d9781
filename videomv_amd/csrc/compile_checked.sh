#!/bin/bash
# compile_checked.sh OUT.o SRC.hip HIPCC FLAGS... — compiles one HIP source with the register report on, keeps the report next to the
# object (OUT.o.res, read by tools/kernel_resources.py) and REFUSES a kernel that spills vector registers: every kernel of the
# production library runs spill-free (a 10-register spill in the epilogue of gemm_xglds<2,5> sat on 18 % of the step).
out=$1; src=$2; hipcc=$3; shift 3
"$hipcc" "$@" -Rpass-analysis=kernel-resource-usage -c "$src" -o "$out" 2> "$out.res"; rc=$?
# (show everything but the remarks and the source-context lines clang prints under each of them)
awk '/remark:/{skip=2; next} skip>0 && (/^ +[0-9]+ \| / || /^ +\| +\^/){skip--; next} {skip=0; print}' "$out.res" | grep -v "^$" >&2
[ $rc -ne 0 ] && exit $rc
if grep -q "VGPRs Spill: [1-9]" "$out.res"; then
    echo "error: $src: kernels with spilled VGPRs (fix them, or build the experiment sources with EXPERIMENTS=1 ALLOW_SPILLS=1):" >&2
    awk '/Function Name:/{name=$5} /VGPRs Spill: [1-9]/{print "    " name, $0}' "$out.res" | sed 's/\[-Rpass-analysis=kernel-resource-usage\]//; s/^\(.*\) .*remark: */\1  /' >&2
    [ -z "$ALLOW_SPILLS" ] && { rm -f "$out"; exit 1; }
fi
exit 0
