export TMPDIR=/tmp
VARIANTS=44,50,51 timeout 90 python tools/exciter_variants.py 2>&1 | grep -v amdgpu.ids
