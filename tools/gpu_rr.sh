D=gpurun_out/$1; mkdir -p $D
timeout -k 10 400 python -m pytest tests/test_gpu_lstm_rr.py -x -q > $D/tests.log 2>&1; echo "pytest rc=$?" >> $D/tests.log; tail -3 $D/tests.log
bash tools/build_variant.sh /tmp/libstamps.so -DNNTK_REC_STAMPS > $D/build.log 2>&1 && NNTK_LIB=/tmp/libstamps.so timeout -k 10 200 python tools/rr_stamps.py 2>&1 | tee $D/stamps.log
timeout -k 10 200 python bench.py --no-cpu-baseline > $D/bench.json 2> $D/bench.err; python - <<PY
import json
d=json.load(open("$D/bench.json"))
print(d["value"], d["ms_per_step"], d["phase_ms"])
PY
