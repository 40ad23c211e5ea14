"""apply path on the GPU: hipGraph-captured eval forward -> per-video pickles in the reference's
format, checked against the golden produced by the reference's own loop body and writer."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from oracle import vpd_oracle as O
from vpd_amd.load import group_by_frame

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_apply_loop_pickles_match_reference_golden(tmp_path):
    from vpd_amd.apply import embed_dataset, write_embeddings
    from vpd_amd.io import load_pickle
    from vpd_amd.models.rgb import RGBF_EmbeddingModel
    g = np.load(os.path.join(REPO, "tests", "golden", "format_case.npz"))
    arch, D, c_in = "resnet18", 32, 5
    enc = RGBF_EmbeddingModel(arch, D, True, "cuda")
    enc.load_state_dict(O.procedural_state_dict(O.encoder_schema(arch, c_in, D), 5))
    videos = ["vidA", "vidB", "vidC"]
    tasks = g["tasks"]
    for k, tag in ((2, "k2"), (1, "k1")):
        imgs = O.synthetic_crops(len(tasks) * k, c_in, 64, 21).reshape(len(tasks), k, c_in, 64, 64)
        batches = [{"video": torch.tensor(tasks[s:s + 5, 0]), "frame": torch.tensor(tasks[s:s + 5, 1]),
                    "img": imgs[s:s + 5]} for s in range(0, len(tasks), 5)]       # 5 + 5 + 2: tail batch
        all_embs = embed_dataset(enc, batches, len(videos))
        out = tmp_path / tag
        write_embeddings(str(out), videos, all_embs)
        for v in videos:
            embs = load_pickle(str(out / ("%s.emb.pkl" % v)))
            assert [t[0] for t in embs] == sorted(t[0] for t in embs)
            assert all(isinstance(t[0], int) and t[1].dtype == np.float32 and t[2] == {} for t in embs)
            assert embs[0][1].shape == ((2, D) if k == 2 else (D,))
            dense, mask = group_by_frame(embs)
            ref = g["dense/%s/%s" % (tag, v)]
            assert np.array_equal(mask, g["mask/%s/%s" % (tag, v)])
            rel = np.linalg.norm(dense - ref) / np.linalg.norm(ref)
            assert rel <= 2e-2, rel            # bf16 student vs the fp32 reference


def test_streaming_writer_equals_batch_writer(tmp_path):
    """Row f4: pickles flushed as each video completes == pickles written at the end (same tuples, same order)."""
    from vpd_amd.apply import StreamingWriter, embed_dataset, write_embeddings
    from vpd_amd.io import load_pickle
    from vpd_amd.models.rgb import RGBF_EmbeddingModel
    enc = RGBF_EmbeddingModel("resnet18", 32, True, "cuda")
    enc.load_state_dict(O.procedural_state_dict(O.encoder_schema("resnet18", 5, 32), 5))
    videos = ["v0", "v1", "v2", "empty"]
    rs = np.random.RandomState(0)
    tasks = [(v, f) for v, nf in ((0, 7), (1, 4), (2, 9)) for f in rs.permutation(nf)]      # frames out of order
    imgs = O.synthetic_crops(len(tasks) * 2, 5, 64, 22).reshape(len(tasks), 2, 5, 64, 64)
    mk = lambda: [{"video": torch.tensor([t[0] for t in tasks[s:s + 6]]), "frame": torch.tensor([int(t[1]) for t in tasks[s:s + 6]]),
                   "img": imgs[s:s + 6]} for s in range(0, len(tasks), 6)]
    write_embeddings(str(tmp_path / "a"), videos, embed_dataset(enc, mk(), len(videos)))
    w = StreamingWriter(str(tmp_path / "b"), videos, [7, 4, 9, 0])
    assert embed_dataset(enc, mk(), len(videos), writer=w) is None
    assert w.written == ["v0", "v1", "v2"] and all(len(p) == 0 for p in w.pending)      # flushed in completion order
    assert sorted(os.listdir(tmp_path / "a")) == sorted(os.listdir(tmp_path / "b")) == ["v0.emb.pkl", "v1.emb.pkl", "v2.emb.pkl"]
    for v in ("v0", "v1", "v2"):
        a, b = load_pickle(str(tmp_path / "a" / (v + ".emb.pkl"))), load_pickle(str(tmp_path / "b" / (v + ".emb.pkl")))
        assert [t[0] for t in a] == [t[0] for t in b] == sorted(t[0] for t in a)
        assert all(np.array_equal(x[1], y[1]) and x[2] == y[2] == {} for x, y in zip(a, b))


def test_train_cli_synthetic_writes_reference_files(tmp_path):
    save = tmp_path / "run"
    r = subprocess.run([sys.executable, os.path.join(REPO, "train_vpd_model.py"), "diving48", "--save_dir", str(save),
                        "--num_epochs", "2", "--batch_size", "16", "--flow_img", "flow", "--motion",
                        "--encoder_arch", "resnet18", "--img_dim", "64", "--synthetic", "48", "--synthetic_emb_dim", "16",
                        # fixed crops (no random augmentation): the loss of the second epoch is then below the first's; the
                        # default recipe (augmentation on) has its own test below
                        "--no_augment",
                        "--checkpoint_frequency", "1"], cwd=REPO, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    cfg = json.load(open(save / "config.json"))
    assert cfg["motion"] is True and cfg["embed_time"] is True and cfg["emb_dim"] == 16 and cfg["use_flow"] is True
    loss = json.load(open(save / "loss.json"))
    assert len(loss) == 2 and set(loss[0]) == {"epoch", "train", "val", "dataset_train", "dataset_val"}
    assert loss[1]["train"] < loss[0]["train"]
    for name in ("best_epoch", "epoch0001", "epoch0002"):
        sd = torch.load(save / ("%s.encoder.pt" % name), map_location="cpu")
        assert list(sd.keys()) == list(O.encoder_schema("resnet18", 5, 16).keys())
        dd = torch.load(save / ("%s.decoder.pt" % name), map_location="cpu")
        assert list(dd.keys()) == list(O.decoder_schema(16).keys())
    # an existing save_dir is an error, as in the reference
    r = subprocess.run([sys.executable, os.path.join(REPO, "train_vpd_model.py"), "diving48", "--save_dir", str(save),
                        "--synthetic", "8", "--num_epochs", "1"], cwd=REPO, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "FileExistsError" in r.stderr


def test_train_cli_default_flags_run_the_reference_recipe_on_the_device(tmp_path, monkeypatch):
    """with the reference's own flags (no extra switch) every train AND val batch goes through the device
    input pipeline (vpd_plan_stage_crops: ColorJitter, mask noise, RandomResizedCrop, flip), as the reference's datasets
    always augment (vpd_dataset/common.py:85-92, single_frame.py:267-272); --no_augment is the only opt-out."""
    import train_vpd_model
    from vpd_amd.engine import StudentEngine
    calls = {"stage": 0, "train": 0, "eval": 0}
    orig = StudentEngine.stage_crops

    def counting(self, *a, **k):
        calls["stage"] += 1
        calls["train" if a[9] else "eval"] += 1
        return orig(self, *a, **k)
    monkeypatch.setattr(StudentEngine, "stage_crops", counting)
    monkeypatch.setattr(sys, "argv", ["train_vpd_model.py", "diving48", "--save_dir", str(tmp_path / "a"), "--num_epochs", "1",
                                      "--batch_size", "8", "--flow_img", "flow", "--encoder_arch", "resnet18",
                                      "--img_dim", "64", "--synthetic", "24", "--synthetic_emb_dim", "16"])
    train_vpd_model.main(**vars(train_vpd_model.get_args()))
    assert calls["train"] == 3 and calls["eval"] == 1 and calls["stage"] == 4, calls
    cfg = json.load(open(tmp_path / "a" / "config.json"))
    assert cfg["augment"].startswith("device") and cfg["use_flow"] is True
    calls.update(stage=0, train=0, eval=0)
    monkeypatch.setattr(sys, "argv", sys.argv[:3] + [str(tmp_path / "b")] + sys.argv[4:] + ["--no_augment"])
    train_vpd_model.main(**vars(train_vpd_model.get_args()))
    assert calls["stage"] == 0
    assert json.load(open(tmp_path / "b" / "config.json"))["augment"].startswith("cpu")


def test_u8_apply_path_matches_reference_golden_and_the_fp32_views(tmp_path):
    """decoded u8 frames cross PCIe and the views are built on the device (normalise, flow decode, h-flip
    with x-flow negation) straight into the stem's staging buffer.
    * k = 1: the frames behind the reference-written golden pickles (tests/golden/format, k1) through the u8 path give
      the golden embeddings within EMB_TOL (the reference's loop: apply_vpd_model.py:146-178);
    * k = 2: the pair [orig, h-flip] of every frame equals, bit for bit, what the fp32 path gives for FrameDataset's own
      views of the same frame in the same order (vpd_dataset/single_frame.py:377-400: flip = torch.flip(img, (2,)), flow
      flipped with channel 0 negated)."""
    from vpd_amd.apply import StreamingWriter, embed_dataset
    from vpd_amd.augment import CropAugmenter
    from vpd_amd.io import load_pickle
    from vpd_amd.models.rgb import RGBF_EmbeddingModel
    g = np.load(os.path.join(REPO, "tests", "golden", "format_case.npz"))
    arch, D, c_in, hw = "resnet18", 32, 5, 64
    enc = RGBF_EmbeddingModel(arch, D, True, "cuda")
    enc.load_state_dict(O.procedural_state_dict(O.encoder_schema(arch, c_in, D), 5))
    aug = CropAugmenter("cuda", O.DIVING48_MEAN_STD, hw, True)
    videos = ["vidA", "vidB", "vidC"]
    tasks = g["tasks"]
    rgb, flow = O.synthetic_crops_u8(len(tasks), c_in, hw, 21)
    # the u8 frames really are what the golden's float crops were made of
    ref_f = O.synthetic_crops(len(tasks), c_in, hw, 21)
    mk = lambda: [{"video": torch.tensor(tasks[s:s + 5, 0]), "frame": torch.tensor(tasks[s:s + 5, 1]),
                   "rgb_u8": rgb[s:s + 5], "flow_u8": flow[s:s + 5]} for s in range(0, len(tasks), 5)]
    fpv = [0, 0, 0]
    for v, _ in tasks:
        fpv[v] += 1
    out = tmp_path / "k1"
    embed_dataset(enc, mk(), len(videos), writer=StreamingWriter(str(out), videos, fpv), augmenter=aug, flip=False)
    for v in videos:
        embs = load_pickle(str(out / ("%s.emb.pkl" % v)))
        assert all(isinstance(t[0], int) and t[1].dtype == np.float32 and t[1].shape == (D,) and t[2] == {} for t in embs)
        dense, mask = group_by_frame(embs)
        ref = g["dense/k1/%s" % v]
        assert np.array_equal(mask, g["mask/k1/%s" % v])
        assert np.linalg.norm(dense - ref) / np.linalg.norm(ref) <= 2e-2
    # k = 2 against the fp32 views of the same frames
    flipped = torch.flip(ref_f, (3,)).clone()
    flipped[:, 3, :, :] *= -1                                     # x-flow of the mirrored view
    views = torch.stack([ref_f, flipped], dim=1)                  # [n, 2, C, H, W]: [orig, flip]
    fp32 = embed_dataset(enc, [{"video": torch.tensor(tasks[s:s + 5, 0]), "frame": torch.tensor(tasks[s:s + 5, 1]),
                                "img": views[s:s + 5]} for s in range(0, len(tasks), 5)], len(videos))
    u8 = embed_dataset(enc, mk(), len(videos), augmenter=aug, flip=True)
    for a, b in zip(fp32, u8):
        assert len(a) == len(b) > 0
        for (fa, ea, _), (fb, eb, _) in zip(sorted(a, key=lambda t: t[0]), sorted(b, key=lambda t: t[0])):
            assert fa == fb and ea.shape == eb.shape == (2, D)
            assert np.array_equal(ea, eb), float(np.abs(ea - eb).max())
            assert not np.array_equal(eb[0], eb[1])               # the flipped view is a different crop


@pytest.mark.parametrize("c_in", [5, 3])
def test_table_driven_view_staging_equals_the_general_pipeline(c_in, monkeypatch):
    """vpd_plan_stage_views (one table look-up per byte; apply path) against vpd_plan_stage_crops with identity parameters
    (the train-time kernel: per-pixel divisions), k = 1 and k = 2: the embeddings must agree bit for bit -- the tables hold
    the same expressions evaluated for every possible byte (vpd_dataset/common.py:52-69, single_frame.py:377-400)."""
    from vpd_amd.augment import CropAugmenter
    from vpd_amd.models.rgb import RGBF_EmbeddingModel
    hw, n = 128, 37
    enc = RGBF_EmbeddingModel("resnet18", 32, c_in == 5, "cuda")
    enc.reset_parameters(seed=2)
    enc.eval()
    aug = CropAugmenter("cuda", O.DIVING48_MEAN_STD, hw, c_in == 5)
    rgb, flow = O.synthetic_crops_u8(n, 5, hw, 33)
    rgb = rgb.cuda()
    flow = flow.cuda() if c_in == 5 else None
    for flip in (False, True):
        embs = []
        for fast in ("1", "0"):
            monkeypatch.setenv("VPD_FAST_VIEWS", fast)
            staged = aug.stage_views(enc.engine, rgb, flow, flip)
            assert staged == (n * (2 if flip else 1), hw)
            embs.append(enc.engine.forward_eval(None, staged=staged).cpu().numpy().copy())
        assert np.isfinite(embs[0]).all() and np.abs(embs[0]).max() > 0
        assert np.array_equal(embs[0], embs[1]), float(np.abs(embs[0] - embs[1]).max())
