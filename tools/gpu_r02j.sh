#!/usr/bin/env bash
python -m pytest tests/test_gpu_embed.py -m gpu -q -x --timeout 600 2>&1 | tail -4
for t in 0 1; do
  python tools/bench_embed.py --uniform --vocab 4000000 --tune embed_wave_per_line=$t 2>&1 | grep lines
  python tools/bench_embed.py --uniform --tune embed_wave_per_line=$t 2>&1 | grep lines
  python tools/bench_embed.py --tune embed_wave_per_line=$t 2>&1 | grep lines
  python tools/bench_embed.py --max-tok 8 --tune embed_wave_per_line=$t 2>&1 | grep lines
done
