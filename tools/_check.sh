mkdir -p gpurun_out
for v in "" "p1_blocks=64" "p1_blocks=128" "p1_blocks=256" "p1_blocks=512" "p1_blocks=2048"; do
  echo "== $v"
  DC3HIP_DEBUG=$v timeout 200 python3 tools/one_build.py 1073741824:0:2 --builds 4 --check | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['build_ms'], d['sufcheck'], {k:v for k,v in d['phase_ms'].items()})"
done
