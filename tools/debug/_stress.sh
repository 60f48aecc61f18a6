mkdir -p gpurun_out/stress; rm -f gpurun_out/stress/*
for tag in full rot6d; do
  for i in 1 2 3 4 5; do TTK_DETERMINISTIC=1 timeout 800 python tools/debug/forward_repeat.py $tag 256 200 > gpurun_out/stress/${tag}_$i.txt 2>&1 & done
  wait
  echo "== $tag"; grep -h "iterations,\|Error\|error" gpurun_out/stress/${tag}_*.txt | cut -c1-200; grep -h "first:" gpurun_out/stress/${tag}_*.txt | cut -c1-200 | head -8
done
