mkdir -p gpurun_out
python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -3
for v in "" ; do
  echo "== $v"
  env $v DC3HIP_LEVEL_PHASES=1 python3 tools/one_build.py 1073741824:2 --builds 2 --check 2> gpurun_out/lv.txt | cut -c1-80
  grep "level [0-3]" gpurun_out/lv.txt | tail -4
done
DC3HIP_NO_TEXT_SHORTCUT=1 python3 tools/one_build.py 1073741824:0 --builds 2 --check | cut -c1-100
python3 tools/real_text_probe.py 256 | cut -c1-200
