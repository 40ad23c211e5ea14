"""A few train steps of the BASELINE student on a fixed synthetic batch -> losses + a digest of every parameter and BatchNorm
buffer.  Used to compare two builds / switch settings of the library on the same GPU: `VPD_LIB_PATH=old.so python tools/step_digest.py`
against the tree must print the same line when a change claims to compute the same numbers."""
import argparse
import hashlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--arch", default="resnet34")
    ap.add_argument("--eval-out", default=None, help="also embed the batch in eval mode and save the array here (.npy)")
    ap.add_argument("--eval-first", action="store_true", help="... before the steps (freshly initialised weights) instead of after them")
    args = ap.parse_args()
    from vpd_amd.data import RGB_MEAN_STD
    from vpd_amd.models.rgb import RGBF_EmbeddingModel
    from vpd_amd.models.util import step
    from vpd_amd.trainer import ModelTrainer
    device = torch.device("cuda", 0)
    torch.manual_seed(0)
    enc = RGBF_EmbeddingModel(args.arch, bench.EMB_DIM, True, device, in_channels=5)
    enc.reset_parameters(seed=0)
    trainer = ModelTrainer(enc, motion=False)
    optimizer, scaler = trainer.get_optimizer(5e-4)
    img, emb = bench.synthetic_batch(args.batch, device, seed=1, c_in=5, mean_std=RGB_MEAN_STD["diving48"], target_dim=bench.EMB_DIM)
    if args.eval_out and args.eval_first:
        import numpy as np
        np.save(args.eval_out, enc.embed(img))
    enc.train()
    eng = enc.engine
    losses = []
    for _ in range(args.steps):
        loss = trainer._forward_loss(img, emb, train=True)
        step(optimizer, scaler, loss)
        losses.append(float(eng.loss_step.item()))
    torch.cuda.synchronize()
    h = hashlib.sha256()
    h.update(eng.params.detach().cpu().numpy().tobytes())
    h.update(eng.bn_running.detach().cpu().numpy().tobytes())
    print("losses %s  params+bn sha256 %s" % (" ".join("%.9g" % v for v in losses), h.hexdigest()[:20]))
    if args.eval_out and not args.eval_first:
        import numpy as np
        np.save(args.eval_out, enc.embed(img))


if __name__ == "__main__":
    main()
