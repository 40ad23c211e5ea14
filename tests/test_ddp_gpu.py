"""The bucketed all-reduce path (HIP events recorded by vpd_backward -> side stream -> RCCL) on ONE GPU:
a world_size-1 nccl group makes SUM all-reduce the identity, so gradients and the loss trajectory must equal
the plain single-GPU run bit for bit, while every event / stream hand-off is exercised."""
import os

import numpy as np
import pytest
import torch

from oracle import vpd_oracle as O

pytestmark = pytest.mark.gpu


def test_bucket_reducer_world1_matches_plain_run():
    import torch.distributed as dist
    from vpd_amd.models.rgb import RGBF_EmbeddingModel
    from vpd_amd.trainer import ModelTrainer
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        sd = O.procedural_state_dict(O.encoder_schema("resnet18", 5, 32), 2)
        img = O.synthetic_crops(6, 5, 64, 3)
        tgt = O.synthetic_targets(6, 32, False, 4)
        res = []
        for use_group in (False, True):
            enc = RGBF_EmbeddingModel("resnet18", 32, True, "cuda")
            enc.load_state_dict(sd)
            tr = ModelTrainer(enc, False, process_group=dist.group.WORLD if use_group else None)
            assert (tr._reducer is not None) == use_group
            opt, sc = tr.get_optimizer(5e-4)
            enc.train()
            loss = tr._forward_loss(img, tgt, train=True)
            loss.backward()
            torch.cuda.synchronize()
            g = enc.engine.grads.clone()
            opt.step()
            traj = [tr.epoch([{"img": img, "emb": tgt}], optimizer=opt, scaler=sc) for _ in range(2)]
            res.append((g.cpu().numpy(), traj))
        # wgrad uses fp32 atomics (summation order varies run to run): tight tolerance, not bitwise
        rel = np.linalg.norm(res[0][0] - res[1][0]) / np.linalg.norm(res[0][0])
        assert rel < 1e-5, rel
        assert np.allclose(res[0][1], res[1][1], rtol=1e-3)
    finally:
        dist.destroy_process_group()
