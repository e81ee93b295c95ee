cp semtools_amd/lib/libsemtools_hip.so /tmp/orig.so
for e in 256 257 263; do
  cp tools/exp_libs/libsemtools_hip_exp$e.so semtools_amd/lib/libsemtools_hip.so
  timeout 200 python tools/trace_k3.py > /tmp/t.out 2>/dev/null; python -c "
import json; d=json.loads(open('/tmp/t.out').read().strip().splitlines()[-1]); print('exp $e clock GHz', d['shader_clock_GHz_over_8_steps'], 'step ticks', d['waves']['0']['total_ticks'] if '0' in d['waves'] else None)"
done
cp /tmp/orig.so semtools_amd/lib/libsemtools_hip.so
