#!/bin/bash
# What one rank's share of the strong-scaling job (64 frames over N GPUs) costs on one GPU: 64 / 32 / 16 / 8 frames per step.
#   bash tools/share_sweep.sh > gpurun_out/share.log     (on the GPU box)
for n in 64 32 16 8; do
  python bench.py --frames $n --extras "" --cpu-frames 0 --steps 30 --warmup 20 2>/dev/null | python -c "
import sys, json
b = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
k = b['kernel_ms_per_step']
print('$n frames: %.3f ms per step, %.4f ms per frame | integral %.3f | launches %s' % (b['ms_per_step'], b['ms_per_step'] / $n, k['integral'], ' '.join('%s%s:%.2f' % (l['kind'], l['stages'], l['ms']) for l in k['launches'])))"
done
