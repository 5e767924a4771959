# 4-wave (one hop) against 8-wave (two hops) oscillator workgroups beside live recurrences: bit-equality, then same-box
# A/B of the one-stream kernel, the pipelined step and the realistic-input step
export TMPDIR=/tmp
mkdir -p gpurun_out/hpb
python - <<'PY'
import os, sys, torch, importlib, subprocess
sys.path.insert(0, os.getcwd())
code = '''
import os, sys, torch, importlib
sys.path.insert(0, os.getcwd())
nws = importlib.import_module("neural-waveshaping-synthesis_amd"); nws.ensure_default_config()
m = nws.NeuralWaveshaping.load_from_checkpoint("tests/golden/weights_vn.npz").cuda().eval(); m.newt = nws.FastNEWT(m.newt)
g = torch.Generator(device="cuda").manual_seed(3)
f0 = 100 + 700 * torch.rand(5, 1, 77, device="cuda", generator=g); c = torch.randn(5, 2, 77, device="cuda", generator=g)
pu = torch.rand(101, device="cuda", generator=g); nz = torch.rand(128 * 77 - 1, device="cuda", generator=g)
with torch.no_grad(): y = m(f0, c, phase_u=pu, noise=nz)
torch.save(y.cpu(), sys.argv[1])
'''
for h in ("1", "2"):
    subprocess.run([sys.executable, "-c", code, f"gpurun_out/hpb/y{h}.pt"], env=dict(os.environ, NWS_EXCITER_HPB=h), check=True)
a, b = torch.load("gpurun_out/hpb/y1.pt"), torch.load("gpurun_out/hpb/y2.pt")
print("one-hop workgroups bit-equal to two-hop:", bool(torch.equal(a, b)), float((a - b).abs().max()))
PY
Q="--no-cpu-baseline --pmc off --legs 0 --batch1-iters 0 --steps 200"
for i in 1 2; do
  for h in 2 1; do
    NWS_EXCITER_HPB=$h timeout 120 python bench.py $Q > gpurun_out/hpb/pipe_$h.json 2>/dev/null
    NWS_EXCITER_HPB=$h timeout 120 python bench.py $Q --inputs realistic > gpurun_out/hpb/real_$h.json 2>/dev/null
    python - <<PY
import json
p=json.loads(open('gpurun_out/hpb/pipe_$h.json').read().strip().splitlines()[-1]); r=json.loads(open('gpurun_out/hpb/real_$h.json').read().strip().splitlines()[-1])
print('hops per workgroup', $h, 'one-stream exciter ms', p['stage_ms']['exciter_newt'], 'pipelined ms/step', round(p['ms_per_step'], 4), '| realistic: one-stream', r['stage_ms']['exciter_newt'], 'ms/step', round(r['ms_per_step'], 4))
PY
  done
done
