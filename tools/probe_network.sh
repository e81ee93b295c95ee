#!/usr/bin/env bash
# Does the GPU lease reach crates.io / PyPI / rustup?  
# Writes gpurun_out/net_probe.txt; every probe is bounded by its own timeout.
out="$(pwd)/gpurun_out"; mkdir -p "$out"; f="$out/net_probe.txt"
{
  echo "# network probe from the GPU lease, $(date -u +%FT%TZ)"
  echo "## toolchains"; for t in cargo rustc rustup go node javac; do printf '%s: ' $t; (command -v $t && $t --version 2>&1 | head -1) || echo absent; done
  echo "## DNS"; for h in crates.io static.crates.io index.crates.io pypi.org files.pythonhosted.org sh.rustup.rs github.com huggingface.co; do printf '%s: ' $h; timeout 8 getent hosts $h || echo "no resolution (rc $?)"; done
  echo "## HTTPS"; for u in https://index.crates.io/config.json https://pypi.org/simple/simsimd/ https://sh.rustup.rs https://github.com https://huggingface.co/minishlab/potion-multilingual-128M/resolve/main/config.json; do printf '%s: ' $u; timeout 15 curl -sS -o /dev/null -w '%{http_code}\n' --max-time 12 "$u" 2>&1 | tail -1; done
  echo "## pip"; cd /tmp; timeout 40 python -m pip download --no-deps -d /tmp/pipdl simsimd==6.5.1 2>&1 | tail -3
  timeout 40 python -m pip download --no-deps -d /tmp/pipdl model2vec 2>&1 | tail -3
  echo "## python twins importable?"; python - <<'PY'
for m in ("simsimd", "model2vec", "qdrant_client", "tokenizers", "safetensors"):
    try:
        mod = __import__(m); print(m, "yes", getattr(mod, "__version__", ""))
    except Exception as e:
        print(m, "no:", type(e).__name__)
PY
  echo "## env proxies"; env | grep -i -E '^(https?|no)_proxy=' || echo none
  echo "## routes"; (ip route 2>/dev/null || cat /proc/net/route) | head -5
} > "$f" 2>&1
cat "$f"
