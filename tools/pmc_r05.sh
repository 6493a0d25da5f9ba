set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_r05
rm -rf $O && mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --no-frames --no-cpu-baseline --no-cfg5 --single-stream --steps 20 --warmup 5"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -o p -- $B > $O/fetch.log 2>&1
echo fetch done
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -o p -- $B > $O/write.log 2>&1
echo write done
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/sq -o p -- $B > $O/sq.log 2>&1
echo sq done
cd $R && python3 tools/pmc_match.py $O > $O/pmc_match.json && python3 -c "
import json; d=json.load(open('$O/pmc_match.json')); print(d['hamming_knn2_kernel_per_launch']); print(d.get('valu')); print(d['grid_sizes_seen'])"
find $O -name "*counter_collection.csv" -delete
