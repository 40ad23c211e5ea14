"""Device input pipeline (row f1) through the C ABI against its CPU oracle, on seeded u8 crops.

Tolerance: the pipeline is fp32 elementwise arithmetic in torchvision's operation order (FMA contraction is
switched off in augment.hip); what is left is the summation order of the contrast op's grey mean (a 16384-term
fp32 sum) -> 2e-5 absolute on values of magnitude <= ~6."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ATOL = 2e-5
MEAN_STD = ((0.3411329922282787, 0.46349889258964044, 0.5162481674015696),
            (0.16302619019820488, 0.17092395707914718, 0.19266662199338647))


def _crops(n, h, w, seed):
    rs = np.random.RandomState(seed)
    # smooth-ish images with flat regions (r == g == b ties and saturated pixels exercise the hue branches)
    rgb = rs.randint(0, 256, (n, h, w, 3)).astype(np.uint8)
    rgb[:, : h // 4, : w // 4, :] = rgb[:, : h // 4, : w // 4, :1]           # grey block
    rgb[:, h // 2:, : w // 8, :] = 255                                          # white block
    rgb[:, : h // 8, w // 2:, :] = 0                                            # black block
    flow = np.clip(np.round(124 + 12 * rs.randn(n, h, w, 2)), 0, 255).astype(np.uint8)
    mask = ((rs.rand(n, h, w) > 0.4) * 255).astype(np.uint8)
    noise = rs.randn(n, 3, h, w).astype(np.float32)
    return rgb, flow, mask, noise


def _oracle_batch(rgb, flow, mask, noise, params, out_dim):
    from oracle import augment_oracle as AO
    outs = []
    for k in range(rgb.shape[0]):
        p = {'order': [int(v) for v in params['order'][k]], 'factors': tuple(float(v) for v in params['factor'][k]),
             'flip': bool(params['flip'][k]), 'noise': bool(params['noise'][k]),
             'crop': tuple(int(v) for v in params['crop'][k])}
        outs.append(AO.augment_item(rgb[k], None if flow is None else flow[k], None if mask is None else mask[k],
                                    None if noise is None else torch.from_numpy(noise[k]), p, MEAN_STD[0], MEAN_STD[1],
                                    out_dim))
    return torch.stack(outs)


@pytest.mark.parametrize("shape", [(6, 128, 128, 128, True), (5, 96, 112, 128, True), (4, 64, 64, 64, False)],
                         ids=["128sq_flow", "96x112_to_128_flow", "64sq_rgb"])
def test_augment_matches_oracle(shape):
    from vpd_amd import augment as A
    n, h, w, out_dim, use_flow = shape
    rgb, flow, mask, noise = _crops(n, h, w, seed=n * 7 + h)
    if not use_flow:
        flow = None
    params = A.sample_params(n, h, w, generator=torch.Generator().manual_seed(h + w))
    params['noise'][0], params['noise'][1] = 1, 0
    params['flip'][0], params['flip'][1] = 1, 0
    params['order'][2] = (3, 1, 0, 2)          # hue and brightness before the contrast mean
    params['order'][3] = (1, -1, -1, 3)        # a partial jitter
    params['crop'][1] = (0, 0, h, w)           # full window (pure resize, or exact copy when h == w == out)
    aug = A.CropAugmenter("cuda:0", MEAN_STD, out_dim, use_flow)
    dev = lambda a: None if a is None else torch.from_numpy(a).cuda()
    got = aug(dev(rgb), dev(flow), dev(mask), params, noise=dev(noise)).cpu()
    exp = _oracle_batch(rgb, flow, mask, noise, params, out_dim)
    assert got.shape == exp.shape == (n, 5 if use_flow else 3, out_dim, out_dim)
    err = (got - exp).abs()
    assert float(err.max()) <= ATOL, "max abs err %.3g at %s" % (float(err.max()), np.unravel_index(int(err.argmax()), err.shape))


def test_no_augmentation_is_bit_exact_loader():
    """identity_params: the device output equals (u8/255 - mean)/std and u8/255 - 0.5 bit for bit."""
    from vpd_amd import augment as A
    n, h = 3, 128
    rgb, flow, _, _ = _crops(n, h, h, seed=1)
    aug = A.CropAugmenter("cuda:0", MEAN_STD, h, True)
    got = aug(torch.from_numpy(rgb).cuda(), torch.from_numpy(flow).cuda(), None, A.identity_params(n, h, h)).cpu()
    exp = _oracle_batch(rgb, flow, None, None, A.identity_params(n, h, h), h)
    assert torch.equal(got, exp)


def test_device_noise_statistics_and_determinism():
    from vpd_amd import augment as A
    n, h = 4, 128
    rgb, flow, mask, _ = _crops(n, h, h, seed=2)
    p = A.identity_params(n, h, h)
    p['noise'] = 1
    p['seed'] = (1234, 5)
    aug = A.CropAugmenter("cuda:0", MEAN_STD, h, True)
    args = (torch.from_numpy(rgb).cuda(), torch.from_numpy(flow).cuda(), torch.from_numpy(mask).cuda())
    a = aug(*args, p).cpu()
    b = aug(*args, p).cpu()
    assert torch.equal(a, b)                                   # counter-based: same key, same noise
    clean = aug(*args, A.identity_params(n, h, h)).cpu()
    d = (a - clean)[:, :3]
    m = (torch.from_numpy(mask) != 0).unsqueeze(1).expand(-1, 3, -1, -1)
    assert torch.all(d[~m] == 0) and torch.equal(a[:, 3:], clean[:, 3:])
    z = d[m] / math.sqrt(0.05)
    assert abs(float(z.mean())) < 0.01 and abs(float(z.std()) - 1.0) < 0.01
    assert abs(float((z ** 4).mean()) - 3.0) < 0.1            # normal kurtosis
    p2 = p.copy()
    p2['seed'] = (1235, 5)
    assert not torch.equal(aug(*args, p2).cpu(), a)


def test_staged_input_equals_fp32_batch_path():
    """vpd_plan_stage_crops + forward(x = NULL) == vpd_augment_crops -> fp32 batch -> forward(x): same embeddings,
    bit for bit in eval mode (both round the same fp32 values to bf16 once)."""
    from vpd_amd import augment as A
    from vpd_amd.models.rgb import RGBF_EmbeddingModel
    n, h = 6, 128
    rgb, flow, mask, noise = _crops(n, h, h, seed=3)
    params = A.sample_params(n, h, h, generator=torch.Generator().manual_seed(9))
    enc = RGBF_EmbeddingModel("resnet18", 32, True, torch.device("cuda:0"))
    enc.reset_parameters(seed=0)
    eng = enc.engine
    aug = A.CropAugmenter("cuda:0", MEAN_STD, h, True)
    dv = [torch.from_numpy(a).cuda() for a in (rgb, flow, mask)]
    nz = torch.from_numpy(noise).cuda()
    img = aug(dv[0], dv[1], dv[2], params, noise=nz)
    tgt = torch.randn(n, 32, device="cuda")
    e1 = eng.forward_eval(img).clone()
    staged = aug.stage(eng, dv[0], dv[1], dv[2], params, train=False, noise=nz)
    e2 = eng.forward_eval(None, staged=staged).clone()
    assert torch.equal(e1, e2)
    enc.train()
    t1 = eng.forward_train(img, tgt, accumulate_loss=False).clone()
    l1 = float(eng.loss_step.item())
    staged = aug.stage(eng, dv[0], dv[1], dv[2], params, train=True, noise=nz)
    t2 = eng.forward_train(None, tgt, accumulate_loss=False, staged=staged).clone()
    # train mode: the batch statistics are accumulated with fp32 atomics, whose order varies from launch to launch
    assert torch.allclose(t1, t2, rtol=1e-3, atol=1e-4) and abs(l1 - float(eng.loss_step.item())) <= 1e-3 * l1


def test_trainer_epoch_on_raw_u8_batches():
    """ModelTrainer.epoch with an augmenter consumes raw u8 batches (device pipeline + staged forward) and, with
    augmentation off, returns the same epoch loss as the fp32 batches of the same crops."""
    from torch.utils.data import DataLoader
    from vpd_amd import augment as A
    from vpd_amd.data import SyntheticCrops
    from vpd_amd.models.rgb import RGBF_EmbeddingModel
    from vpd_amd.trainer import ModelTrainer
    losses = []
    for raw in (False, True):
        enc = RGBF_EmbeddingModel("resnet18", 32, True, torch.device("cuda:0"))
        enc.reset_parameters(seed=0)
        aug = A.CropAugmenter("cuda:0", MEAN_STD, 64, True) if raw else None
        tr = ModelTrainer(enc, motion=False, augmenter=aug, augment=False)
        opt, scaler = tr.get_optimizer(5e-4)
        ds = SyntheticCrops(24, 5, 64, 32, False, MEAN_STD, seed=3, raw_u8=raw)
        losses.append([tr.epoch(DataLoader(ds, batch_size=8), opt, scaler) for _ in range(2)])
    # same crops, same bf16 staging values (bit-exact, test_staged_input_equals_fp32_batch_path); the training steps
    # in between use fp32 atomics in the generic weight-gradient kernel, whose summation order varies run to run
    assert np.allclose(losses[0], losses[1], rtol=1e-3), losses
    # with augmentation on: runs, finite, and differs from the un-augmented loss
    enc = RGBF_EmbeddingModel("resnet18", 32, True, torch.device("cuda:0"))
    enc.reset_parameters(seed=0)
    tr = ModelTrainer(enc, motion=False, augmenter=A.CropAugmenter("cuda:0", MEAN_STD, 64, True), augment=True)
    opt, scaler = tr.get_optimizer(5e-4)
    la = tr.epoch(DataLoader(SyntheticCrops(24, 5, 64, 32, False, MEAN_STD, seed=3, raw_u8=True), batch_size=8), opt, scaler)
    # (an untrained student's loss is dominated by the random targets: the augmentation moves it in the 4th digit)
    assert math.isfinite(la) and abs(la - losses[1][0]) > 1e-6 * losses[1][0]


def test_trainer_draws_a_fresh_noise_key_per_batch():
    """the device mask noise is keyed by a per-BATCH Philox key; the trainer must draw a new one for every
    batch (the same key made every step of every epoch add the same noise pattern to a batch slot)."""
    from vpd_amd.augment import CropAugmenter
    from vpd_amd.models.rgb import RGBF_EmbeddingModel
    from vpd_amd.trainer import ModelTrainer
    n, h = 4, 64
    rgb, flow, _, _ = _crops(n, h, h, seed=5)
    rgb, flow = torch.from_numpy(rgb), torch.from_numpy(flow)
    mask = torch.full((n, h, h), 255, dtype=torch.uint8)
    enc = RGBF_EmbeddingModel("resnet18", 16, True, "cuda")
    aug = CropAugmenter(enc.device, MEAN_STD, h, True)
    tr = ModelTrainer(enc, False, augmenter=aug)
    batch = {"rgb_u8": rgb, "flow_u8": flow, "mask_u8": mask, "emb": torch.zeros(n, 16)}
    keys, staged = [], []
    for _ in range(3):
        tr._forward_loss_raw(batch, train=False)
        keys.append(tuple(int(v) for v in tr.last_aug_params["seed"][0]))
        assert (tr.last_aug_params["seed"] == tr.last_aug_params["seed"][0]).all()
    assert len(set(keys)) == 3, keys
    # and the key reaches the device: identical decisions + different keys -> different noise on the noisy crops
    p = tr.last_aug_params.copy()
    p["noise"] = 1
    a = aug(rgb.cuda(), flow.cuda(), mask.cuda(), p).clone()
    p2 = p.copy()
    p2["seed"] = (p["seed"][0][0] ^ 1, p["seed"][0][1])
    b = aug(rgb.cuda(), flow.cuda(), mask.cuda(), p2)
    assert float((a[:, :3] - b[:, :3]).abs().max()) > 0.05 and torch.equal(a[:, 3:], b[:, 3:])


def test_staging_calls_reject_more_rows_than_a_grid_holds():
    """One grid row per view / crop: beyond gridDim.y = 65535 the staging entry points fail with a message,
    not with an opaque launch error."""
    import ctypes as C
    from vpd_amd._lib import VpdHipError, check, lib
    from vpd_amd.models.rgb import RGBF_EmbeddingModel
    enc = RGBF_EmbeddingModel("resnet18", 32, True, "cuda")
    pl = enc.engine.plan(64, 64, 4, False, False)
    ms = (C.c_float * 6)(*[0.5] * 6)
    one = torch.zeros(16, dtype=torch.uint8, device="cuda")
    p = lambda t: C.c_void_p(t.data_ptr())
    with pytest.raises(VpdHipError, match="65535"):
        check(lib().vpd_plan_stage_views(pl.handle, p(one), p(one), 32768, 2, 64, 64, ms, p(pl.workspace), None), "stage_views")
    # the largest legal row count is refused for the plan's size, not for the grid
    with pytest.raises(VpdHipError) as e:
        check(lib().vpd_plan_stage_views(pl.handle, p(one), p(one), 65535, 1, 64, 64, ms, p(pl.workspace), None), "stage_views")
    assert "65535" not in str(e.value)
