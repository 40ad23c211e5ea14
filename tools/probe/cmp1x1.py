import sys, os, ctypes as C, numpy as np, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import test_ops_gpu as T
g = torch.Generator().manual_seed(3)
n, ci, co, h, w = 64, 64, 256, 32, 32
x = T.bf16_round(torch.randn(n, ci, h, w, generator=g)); wt = T.bf16_round(torch.randn(co, ci, 1, 1, generator=g) * 0.17)
xp = T.to_padded_nhwc(x, 1, 1, 1, 1); taps = T.tapset(1, 1, 1, 1, 1, 1, 0, 1, 1)
y, st = T.run_conv(xp, T.pack_fwd(wt), n, h + 2, w + 2, ci, h, w, 0, h, w, 1, 0, 0, 1, ci, co, taps, want_stats=True)
np.save(sys.argv[1], y.view(torch.int16).cpu().numpy()); np.save(sys.argv[1] + ".st.npy", st.sum(0).cpu().numpy())
