import os, sys, time, ctypes
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/gcn-bmp_amd')
import numpy as np, torch
from bmp import synth, packed, _lib
from bmp.predictor import build_pair_predictor
dev = torch.device('cuda:0')
store = synth.make_store(); ms = packed.MolStore(store)
i1, i2, lab = synth.make_pairs(limit=1024)
pb = packed.pack_from_store(ms, [i1, i2], device=dev)
model = build_pair_predictor(128, 128, 4, attn='nie').to(dev)
t = torch.from_numpy(lab.reshape(-1, 1)).to(dev)
L = _lib.lib()
out = (ctypes.c_double * 3)()
def run():
    y = model(pb); loss = model.loss(y, t); loss.backward()
for _ in range(3): run()
torch.cuda.synchronize()
L.bmp_prof_start(int(os.environ.get("KCLS", "4")))
for _ in range(5): run()
torch.cuda.synchronize()
n = L.bmp_prof_stop(out)
print('stop', os.environ.get('BMP_FZ_STOP'), 'coattn class ms/step', out[0] / 5, 'launches/step', n / 5)
