cd $GRAFT_REPO_ROOT
python tests/checkers/fuzz_parity.py 400 fp32 > gpurun_out/r05_fuzz_full_fp32.txt 2>&1; tail -4 gpurun_out/r05_fuzz_full_fp32.txt > gpurun_out/r05_fuzz_sweep_400_fp32.txt; cat gpurun_out/r05_fuzz_sweep_400_fp32.txt
python tools/stress_determinism.py > gpurun_out/r05_stress_determinism.txt 2>&1; cat gpurun_out/r05_stress_determinism.txt
python - <<'PY' > gpurun_out/r05_small_determinism.txt 2>&1
# 300 repeats of the 32- and 108-atom cells through the small-system path: identical bits every time
import sys, torch
sys.path[:0] = ['.', 'torch-m3gnet_amd', 'tests']
from torch_m3gnet.data.synthetic import fcc_cu_graph
from torch_m3gnet.data import MaterialGraphKey as K
from torch_m3gnet.model.build import build_model
torch.manual_seed(0)
model = build_model(5.0, 4.0, 3, 3, 95, 64, 3).cuda()
for n in (2, 3, 6):
    g = fcc_cu_graph(n, n, n).to('cuda')
    out = model(g)
    ref = {k: out[k].clone() for k in (K.TOTAL_ENERGY, K.FORCES, K.STRESSES)}
    bad = 0
    for i in range(300):
        o = model(g)
        bad += any(not torch.equal(o[k], v) for k, v in ref.items())
    print(f'{4 * n ** 3} atoms: mismatching repeats: {bad} of 300', flush=True)
PY
cat gpurun_out/r05_small_determinism.txt
