"""development: Steger stage time on synthetic frames vs all-dark frames of the same shape"""
import time, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from laser_3d_reconstruction_amd.core.laser_extractor import FastStegerExtractor
from laser_3d_reconstruction_amd.utils import synthetic as S
W, H, B = 640, 360, 256
fr = [S.stereo_pair(W, H, 64, k)[0] for k in range(16)]
lb = np.stack([fr[i % 16] for i in range(B)])
st = FastStegerExtractor(3.0, 200, device=0)
for name, arr in (("synthetic", lb), ("dark", np.zeros_like(lb)), ("bright", np.full_like(lb, 255))):
    d = torch.from_numpy(arr).cuda()
    st.extract_device(d, 4 * H); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(10): xy, c = st.extract_device(d, 4 * H)
    torch.cuda.synchronize()
    print(name, "ms", (time.perf_counter() - t) * 100, "pts/frame", float(c.float().mean()))
