"""The reference's own f32 reproducibility on a config (oracle == reference bit for bit, tests/golden/e2e_*.npz): the same
model and images with 8 CPU threads, 1 thread and in float64.  Where this exceeds 1e-3 px the 1e-3 gate of
BASELINE.json is below the f32 noise floor of the reference itself and the parity tests use the measured floor instead
(yolov8s: 8 vs 1 threads 2.2e-3 px, f32 vs f64 1.8e-3 .. 2.9e-3 px; scores 4e-6).   python tools/ref_noise_floor.py yolov8s"""
import torch, numpy as np, sys
sys.path.insert(0, str(__import__('pathlib').Path(__file__).resolve().parents[1]))
from oracle import tasks as ot
from ultralytics_pro_amd.utils import procedural as P
name=sys.argv[1]
g=np.load(str(__import__('pathlib').Path(__file__).resolve().parents[1] / 'tests' / 'golden' / f'e2e_{name}.npz'))
m=ot.DetectionModel(name+'.yaml'); P.apply_procedural_weights(m); m.fuse()
x=P.synthetic_images(2)
sel=g['anchor_sel']
with torch.no_grad():
    torch.set_num_threads(8); y8=m(x)[0]
    torch.set_num_threads(1); y1=m(x)[0]
    m64=m.double(); y64=m64(x.double())[0]
d=lambda a,b: (float((a[:,:4]-b[:,:4]).abs().max()), float((a[:,4:]-b[:,4:]).abs().max()))
print(name,'8 vs 1 threads',d(y8,y1),'8thr vs f64',d(y8.double(),y64),'1thr vs f64',d(y1.double(),y64))
print('golden vs 8thr', np.abs(y8[:,:,sel].numpy()-g['y_sel']).max(), 'golden vs f64 (sel) box', np.abs(y64[:,:4][:,:,sel].numpy()-g['y_sel'][:,:4]).max())
