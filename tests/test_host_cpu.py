"""Host-side logic that needs no GPU: dataset contracts, file formats, CLI flag surface."""
import json
import os
import pickle
import sys

import numpy as np
import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(REPO, "tests", "golden", "format")


def test_reference_written_pickles_match_the_product_reader():
    """vpd_amd.load.group_by_frame (row f4, reference action_dataset/load.py:16-43) on pickles the reference wrote, against
    the dense arrays the reference's own group_by_frame made of them."""
    g = np.load(os.path.join(REPO, "tests", "golden", "format_case.npz"))
    from vpd_amd.io import load_pickle
    from vpd_amd.load import group_by_frame
    for tag in ("k1", "k2"):
        for vid in ("vidA", "vidB", "vidC"):
            embs = load_pickle(os.path.join(GOLDEN, "%s.%s.emb.pkl" % (vid, tag)))
            assert isinstance(embs, list) and all(isinstance(t, tuple) and len(t) == 3 for t in embs)
            assert [t[0] for t in embs] == sorted(t[0] for t in embs)
            assert all(isinstance(t[0], int) and t[1].dtype == np.float32 and t[2] == {} for t in embs)
            assert embs[0][1].shape == ((2, 32) if tag == "k2" else (32,))
            dense, mask = group_by_frame(embs)
            assert np.array_equal(dense, g["dense/%s/%s" % (tag, vid)]) and np.array_equal(mask, g["mask/%s/%s" % (tag, vid)])


def test_reader_averages_repeated_frames_and_blends_gaps_like_the_reference(capsys):
    """Repeated frame numbers are averaged (load.py:29-32), gaps blended with the reference's weights (:34-42), leading
    frames stay zero, other files of the directory are skipped, norm=True L2-normalises rows and leaves all-zero rows
    alone (:46-64): bit for bit the arrays the reference's load_embs produced (tests/golden/reader_case.npz)."""
    from vpd_amd.load import group_by_frame, load_embs, normalize_rows
    g = np.load(os.path.join(REPO, "tests", "golden", "reader_case.npz"))
    rdir = os.path.join(REPO, "tests", "golden", "reader")
    for norm in (False, True):
        d = load_embs(rdir, norm)
        assert sorted(d) == ["one", "rally_a", "rally_b", "zero_row"]
        for name, (dense, mask) in d.items():
            want = g["dense/%d/%s" % (norm, name)]
            assert dense.dtype == np.float64 and dense.shape == want.shape
            assert np.array_equal(dense, want), (norm, name, np.abs(dense - want).max())
            assert np.array_equal(mask, g["mask/%d/%s" % (norm, name)])
    assert "Loading embs:" in capsys.readouterr().out
    # the properties themselves, on a hand-made list: frame 1 seen twice -> mean; frames 2, 3 blended; frame 0 empty
    e = lambda v: np.full(4, v, np.float32)
    dense, mask = group_by_frame([(1, e(2.0), {}), (1, e(4.0), {}), (4, e(9.0), {})])
    assert mask.tolist() == [False, True, False, False, True]
    assert np.array_equal(dense[0], np.zeros(4)) and np.array_equal(dense[1], e(3.0))
    assert np.allclose(dense[2], 1 / 3 * 3.0 + 2 / 3 * 9.0) and np.allclose(dense[3], 2 / 3 * 3.0 + 1 / 3 * 9.0)
    assert np.array_equal(normalize_rows(np.zeros((2, 3))), np.zeros((2, 3)))


def test_config_and_loss_json_schema():
    cfg = json.load(open(os.path.join(GOLDEN, "config.json")))
    assert set(cfg) == {"num_epochs", "batch_size", "learning_rate", "img_dim", "use_flow", "motion", "emb_dim",
                        "encoder_arch", "rgb_mean_std"}
    loss = json.load(open(os.path.join(GOLDEN, "loss.json")))
    assert set(loss[0]) == {"epoch", "train", "val", "dataset_train", "dataset_val"}


def test_apply_batch_size_rule():
    from vpd_amd.apply import apply_batch_size
    assert apply_batch_size(None, False) == 500       # 500 frames x 2 views
    assert apply_batch_size(None, True) == 1000
    assert apply_batch_size(4, False) == 100


def test_cli_flag_surface_matches_reference():
    sys.path.insert(0, REPO)
    import apply_vpd_model
    import train_vpd_model
    old = sys.argv
    try:
        sys.argv = ["x", "diving48", "--save_dir", "d", "--motion", "--flow_img", "flow", "--encoder_arch", "resnet34",
                    "--checkpoint_frequency", "5", "--model_select_window", "3", "--min_pose_score", "0.4",
                    "--emb_dir", "e", "--batch_size", "256", "--learning_rate", "0.001", "--img_dim", "128",
                    "--num_epochs", "2"]
        a = train_vpd_model.get_args()
        assert a.dataset == "diving48" and a.motion and a.flow_img == "flow" and a.batch_size == 256
        sys.argv = ["x", "diving48", "--save_dir", "d"]
        a = train_vpd_model.get_args()
        assert (a.num_epochs, a.batch_size, a.learning_rate, a.img_dim, a.encoder_arch, a.model_select_window) == \
            (1000, 100, 0.0005, 128, "resnet34", 5)
        sys.argv = ["x", "diving48", "--save_dir", "d", "--emb_dir", "a", "--penn_dir", "b"]
        with pytest.raises(SystemExit):
            train_vpd_model.get_args()
        sys.argv = ["x", "model", "-d", "fs", "-o", "out", "-m", "7", "--no_flip", "--flow_img", "flow"]
        b = apply_vpd_model.get_args()
        assert (b.model_dir, b.dataset, b.out_dir, b.model_epoch, b.no_flip, b.flow_img) == \
            ("model", "fs", "out", 7, True, "flow")
    finally:
        sys.argv = old


def test_teacher_dataset_contract(tmp_path):
    """Teacher pickle ingestion: pose-score filter, embed_time pairing of consecutive frames,
    batch items {'img': f32[5,H,W], 'emb': f32[2D]}, flip negates the x-flow channel."""
    from PIL import Image
    from vpd_amd.data import RGB_MEAN_STD, TeacherEmbDataset
    emb_dir, img_dir = tmp_path / "embs", tmp_path / "crops"
    (img_dir / "vid").mkdir(parents=True)
    emb_dir.mkdir()
    rs = np.random.RandomState(0)
    embs = []
    for fr in (3, 4, 5, 7, 8):
        Image.fromarray(rs.randint(0, 255, (32, 32, 3)).astype(np.uint8)).save(img_dir / "vid" / ("%d.png" % fr))
        Image.fromarray(rs.randint(0, 255, (32, 32, 3)).astype(np.uint8)).save(img_dir / "vid" / ("%d.flow.png" % fr))
        embs.append((fr, rs.randn(2, 8).astype(np.float32), {"kp_score": 0.9 if fr != 8 else 0.1}))
    with open(emb_dir / "vid.emb.pkl", "wb") as fp:
        pickle.dump(embs, fp)
    tr, va, D = TeacherEmbDataset.load_default(str(emb_dir), str(img_dir), 32, True, 50, RGB_MEAN_STD["diving48"],
                                               flow_img_name="flow")
    assert D == 8 and len(tr) == 50 and len(va) == 10
    kept = sorted(x[1] for x in tr.data + va.data)
    assert kept == [4, 5]          # 3: no predecessor; 7: gap; 8: low pose score
    item = tr[0]
    assert item["img"].shape == (5, 32, 32) and item["img"].dtype == torch.float32
    assert item["emb"].shape == (16,) and item["emb"].dtype == torch.float32
    assert float(item["img"][3:].abs().max()) <= 0.5
    # a teacher pickle without any pose score cannot be filtered: NotImplementedError, as vpd_dataset/single_frame.py:44
    with open(emb_dir / "bad.emb.pkl", "wb") as fp:
        pickle.dump([(1, rs.randn(2, 8).astype(np.float32), {})], fp)
    with pytest.raises(NotImplementedError):
        TeacherEmbDataset.load_default(str(emb_dir), str(img_dir), 32, False, 50, RGB_MEAN_STD["diving48"])


def test_split_is_shared_across_ranks_when_seeded(tmp_path):
    """Data-parallel runs give every rank the same split seed (train_vpd_model.py broadcasts rank 0's): identical
    train / val frame sets whatever the directory listing order; 20 % (ceil, as sklearn's train_test_split) validate."""
    from vpd_amd.data import RGB_MEAN_STD, TeacherEmbDataset
    emb_dir = tmp_path / "embs"
    emb_dir.mkdir()
    rs = np.random.RandomState(1)
    for v in range(3):
        embs = [(f, rs.randn(8).astype(np.float32), {"kp_score": 0.9}) for f in range(7)]
        with open(emb_dir / ("v%d.emb.pkl" % v), "wb") as fp:
            pickle.dump(embs, fp)
    load = lambda seed: TeacherEmbDataset.load_default(str(emb_dir), "/nonexistent", 32, False, 100,
                                                       RGB_MEAN_STD["fs"], split_seed=seed)
    key = lambda ds: [x[:2] for x in ds.data]
    (t0, v0, _), (t1, v1, _), (t2, v2, _) = load(1234), load(1234), load(99)
    assert key(t0) == key(t1) and key(v0) == key(v1)
    assert key(v0) != key(v2)
    assert len(v0.data) == 5 and len(t0.data) == 16            # ceil(0.2 * 21)
    assert not set(key(t0)) & set(key(v0))


def test_synthetic_dataset_is_seeded_and_in_range():
    from vpd_amd.data import RGB_MEAN_STD, SyntheticCrops
    ds = SyntheticCrops(10, 5, 64, 16, True, RGB_MEAN_STD["diving48"], seed=3)
    a, b = ds[2], ds[2]
    assert torch.equal(a["img"], b["img"]) and a["img"].shape == (5, 64, 64) and a["emb"].shape == (32,)
    assert float(a["img"][3:].abs().max()) <= 0.5


def test_frame_dataset_jitter_views(tmp_path):
    """apply --jitter: view order [orig, j x jit(orig), j x jit(flip), flip] (reference Appendix B.8), ColorJitter applied
    to the NORMALISED image exactly as the oracle's restatement of torchvision's ops composes it."""
    import numpy as np
    import torch
    from PIL import Image
    from oracle import augment_oracle as AO
    from vpd_amd.data import FrameDataset, color_jitter, load_flow, load_rgb
    rs = np.random.RandomState(0)
    d = tmp_path / "vid"
    d.mkdir()
    Image.fromarray(rs.randint(0, 256, (32, 32, 3)).astype(np.uint8)).save(d / "7.png")
    Image.fromarray(rs.randint(0, 256, (32, 32, 3)).astype(np.uint8)).save(d / "7.flow.png")
    mean_std = ((0.34, 0.46, 0.52), (0.16, 0.17, 0.19))
    base = load_rgb(str(d / "7.png"), 32, mean_std)
    flow = load_flow(str(d / "7.flow.png"), 32)
    # color_jitter == the oracle's ops in the sampled order with the sampled factors (same RNG draws)
    torch.manual_seed(5)
    got = color_jitter(base)
    torch.manual_seed(5)
    order, f = AO.color_jitter_params()
    exp = base
    for op in order:
        exp = AO._OPS[op](exp, f[op])
    assert torch.allclose(got, exp, atol=1e-6) and float(got.min()) >= 0.0 and float(got.max()) <= 1.0
    ds = FrameDataset([(0, 7, str(d / "7"))], 32, mean_std, augment_jitter=2, augment_flip=True, flow_img_name="flow")
    item = ds[0]
    v = item["img"]
    assert v.shape == (6, 5, 32, 32) and item["frame"] == 7
    assert torch.equal(v[0, :3], base) and torch.equal(v[0, 3:], flow)
    assert torch.equal(v[5, :3], torch.flip(base, (2,)))                       # the plain flip is LAST
    assert torch.equal(v[5, 3], -torch.flip(flow[0], (1,))) and torch.equal(v[5, 4], torch.flip(flow[1], (1,)))
    for k in (1, 2, 3, 4):                                                     # jittered views: clamped to [0, 1], unflipped flow
        assert float(v[k, :3].min()) >= 0.0 and float(v[k, :3].max()) <= 1.0 and torch.equal(v[k, 3:], flow)
    plain = FrameDataset([(0, 7, str(d / "7"))], 32, mean_std, augment_flip=False)[0]["img"]
    assert plain.shape == (1, 3, 32, 32)


def test_tennis_crop_listing(tmp_path):
    """apply_vpd_model.get_tennis_dataset: two pseudo-videos per clip, frame numbers relative to the clip start."""
    from vpd_amd.data import list_tennis_crops
    vd, cd = tmp_path / "videos", tmp_path / "crops"
    vd.mkdir()
    (vd / "match_a_100_103.mp4").write_bytes(b"")
    (vd / "notes.txt").write_bytes(b"")
    for player, frames in (("front", (100, 102, 103)), ("back", ())):
        d = cd / "match_a" / player
        d.mkdir(parents=True)
        for f in frames:
            (d / ("%d.png" % f)).write_bytes(b"")
        (d / "999.png").write_bytes(b"")            # outside the clip
    videos, tasks = list_tennis_crops(str(vd), str(cd))
    assert videos == ["front__match_a_100_103", "back__match_a_100_103"]
    assert [(v, f) for v, f, _ in tasks] == [(0, 0), (0, 2), (0, 3)]
    assert tasks[1][2].endswith("match_a/front/102")


def test_tennis_teacher_ingestion(tmp_path):
    """TennisDataset.load_default: <player>__<video>_<start>_<end>.emb.pkl -> items addressing
    <video>/<player>/<start + frame>.png, pose-score filter (dp_score before kp_score), embed_time pairing."""
    import numpy as np
    from vpd_amd.data import load_tennis_default
    from vpd_amd.io import store_pickle
    emb_dir = tmp_path / "embs"
    emb_dir.mkdir()
    rs = np.random.RandomState(0)
    for clip in range(5):
        embs = [(f, rs.randn(2, 8).astype(np.float32), {"kp_score": 0.9, "dp_score": (0.1 if f == 2 else 0.8)})
                for f in (0, 1, 2, 3, 5)]
        store_pickle(str(emb_dir / ("front__m%d_100_105.emb.pkl" % clip)), embs)
    tr, va, d = load_tennis_default(str(emb_dir), "/crops", 128, True, 1000, ((0,) * 3, (1,) * 3), flow_img_name="flow")
    assert d == 8 and len(tr) == 1000 and len(va) == 200
    items = tr.data + va.data
    assert len(items) == 5 * 2                                  # per clip: frames 1 and 3 (0 first, 2 filtered, 5 no predecessor)
    assert {it[1] for it in items} == {101, 103} and all(it[0].endswith("/front") for it in items)
    assert all(it[2].shape == (2, 16) for it in items)          # [emb, emb - prev] along the feature axis
    assert len({it[0] for it in tr.data} & {it[0] for it in va.data}) == 0      # split over clips


def test_wgrad128_schedule_covers_every_task_once_and_balances():
    """Host scheduler of conv_wgrad128_persistent_kernel (no GPU): every (problem, tile, split) appears exactly once, the
    splits cover all 64-pixel chunks, block lists are contiguous, and the LPT deal is within Graham's 4/3 bound of the
    trivial lower bounds.  Shapes: the ResNet-34 launches of the 256-crop step (layer2; layer3 + layer4 together) and a
    ragged mix on an odd number of compute units."""
    import ctypes as C
    from vpd_amd._lib import lib
    L = lib()
    cases = {
        "layer2": ([(256 * 256, 128, 128, 108)] * 7, 256),
        "layer3+4": ([(256 * 64, 256, 256, 100)] * 11 + [(256 * 16, 512, 512, 144)] * 5, 256),
        "ragged": ([(5 * 64 + 16, 256, 128, 100), (77 * 256, 128, 64, 108), (3 * 16, 512, 512, 144)], 37),
    }
    for name, (probs, G) in cases.items():
        n = len(probs)
        dims = (C.c_int * (4 * n))(*[v for p in probs for v in p])
        ks = (C.c_int * n)()
        bb = (C.c_int * (G + 1))()
        cap = 1 << 15
        tk = (C.c_int * (4 * cap))()
        est = C.c_double()
        nt = L.vpd_op_wgrad128_schedule(n, dims, G, ks, bb, tk, cap, C.byref(est))
        assert nt > 0, name
        assert bb[0] == 0 and bb[G] == nt and all(bb[b] <= bb[b + 1] for b in range(G)), name
        seen = set()
        load = [0.0] * G
        lens = []
        for b in range(G):
            for t in range(bb[b], bb[b + 1]):
                pi, tile, split = tk[4 * t], tk[4 * t + 1], tk[4 * t + 2]
                M, co, ci, _ = probs[pi]
                nch = (M + 63) // 64
                cpb = (nch + ks[pi] - 1) // ks[pi]
                assert 0 <= tile < (co // 128) * (ci // 64) and 0 <= split < ks[pi], name
                assert split * cpb < nch, (name, "empty split")
                assert (pi, tile, split) not in seen, name
                seen.add((pi, tile, split))
                ln = min(cpb, nch - split * cpb)
                load[b] += ln
                lens.append(ln)
        want = sum((co // 128) * (ci // 64) * ks[i] for i, (M, co, ci, _) in enumerate(probs))
        assert len(seen) == nt == want, name
        for i, (M, co, ci, _) in enumerate(probs):      # every chunk of every tile is covered by exactly one split
            nch = (M + 63) // 64
            cpb = (nch + ks[i] - 1) // ks[i]
            assert (ks[i] - 1) * cpb < nch <= ks[i] * cpb, name
        lower = max(sum(lens) / G, max(lens))
        assert max(load) <= 4.0 / 3.0 * lower + max(lens) * 0.34 + 1e-9, (name, max(load), lower)
        assert est.value > 0


def test_frame_dataset_u8_items_equal_the_float_items(tmp_path):
    """FrameDataset(raw_u8=True) reads the same PNGs as the float path: decoding its u8 item the way the reference does
    (vpd_dataset/common.py:52-69: /255, normalise; flow channels as cv2 reads them, /255 - 0.5) gives the float item."""
    from PIL import Image
    from vpd_amd.data import FrameDataset, RGB_MEAN_STD
    rs = np.random.RandomState(3)
    d = tmp_path / "crops" / "vid0"
    d.mkdir(parents=True)
    tasks = []
    for f in range(3):
        Image.fromarray(rs.randint(0, 256, (64, 64, 3)).astype(np.uint8)).save(str(d / ("%d.png" % f)))
        Image.fromarray(rs.randint(0, 256, (64, 64, 3)).astype(np.uint8)).save(str(d / ("%d.raft.png" % f)))
        tasks.append((0, f, str(d / str(f))))
    ms = RGB_MEAN_STD["diving48"]
    fl = FrameDataset(tasks, 64, ms, augment_flip=True, flow_img_name="raft")
    u8 = FrameDataset(tasks, 64, ms, augment_flip=True, flow_img_name="raft", raw_u8=True)
    mean = torch.tensor(ms[0]).view(3, 1, 1)
    std = torch.tensor(ms[1]).view(3, 1, 1)
    for i in range(3):
        a, b = fl[i], u8[i]
        assert (a["video"], a["frame"]) == (b["video"], b["frame"])
        assert b["rgb_u8"].dtype == torch.uint8 and tuple(b["rgb_u8"].shape) == (64, 64, 3) and tuple(b["flow_u8"].shape) == (64, 64, 2)
        rgb = (b["rgb_u8"].permute(2, 0, 1).float() / 255. - mean) / std
        flow = (b["flow_u8"].permute(2, 0, 1).double() / 255. - 0.5).float()
        assert torch.equal(torch.cat([rgb, flow]), a["img"][0])
    with pytest.raises(ValueError):
        FrameDataset(tasks, 64, ms, augment_jitter=2, raw_u8=True)


def test_no_test_video_holds_the_reference_test_split_out(tmp_path, monkeypatch, capsys):
    """--no_test_video (train_vpd_model.py:125-156): the prefixes are the reference's (action_dataset/eval.py:3-43) and the
    loaders drop every teacher pickle that starts with one of them; fx / diving48 derive theirs from the label files."""
    sys.path.insert(0, REPO)
    import train_vpd_model
    from vpd_amd.data import RGB_MEAN_STD
    from vpd_amd.splits import FS_TEST_PREFIXES, get_test_prefixes
    assert FS_TEST_PREFIXES == ("men_olympic_short_program_2018", "men_world_short_program_2018",
                                "women_olympic_short_program_2018", "women_world_short_program_2018")
    t = get_test_prefixes("tennis")
    assert len(t) == 12 and "front__usopen_2019_womens_osaka_gauff" in t and "wimbledon_2019_mens_semifinal_federer_nadal" in t
    emb_dir = tmp_path / "embs"
    emb_dir.mkdir()
    rs = np.random.RandomState(0)
    for name in ("men_world_short_program_2018_03", "men_world_short_program_2017_03"):
        with open(emb_dir / (name + ".emb.pkl"), "wb") as fp:
            pickle.dump([(f, rs.randn(2, 8).astype(np.float32), {"kp_score": 0.9}) for f in range(6)], fp)
    kw = {"img_dim": 32, "flow_img_name": None, "embed_time": False, "rgb_mean_std": RGB_MEAN_STD["fs"], "target_len": 10,
          "split_seed": 1}
    tr, va, D = train_vpd_model.load_dataset("fs", dict(kw), str(emb_dir), None, True)
    assert {x[0] for x in tr.data + va.data} == {"men_world_short_program_2017_03"}
    assert "Excluded: men_world_short_program_2018_03" in capsys.readouterr().out
    tr, va, D = train_vpd_model.load_dataset("fs", dict(kw), str(emb_dir), None, False)
    assert len({x[0] for x in tr.data + va.data}) == 2
    # label-file driven splits
    gym = tmp_path / "gym99_val_element.txt"
    gym.write_text("vidA_E_000100_000200_A_0010_0020 5\nvidB_E_000300_000400_A_0001_0002 7\n")
    monkeypatch.setenv("VPD_GYM99_VAL_FILE", str(gym))
    assert get_test_prefixes("fx") == ("vidA_E_000100_000200", "vidB_E_000300_000400")
    dv = tmp_path / "Diving48_V2_test.json"
    dv.write_text(json.dumps([{"vid_name": "a", "start_frame": 0, "end_frame": 5}, {"vid_name": "b", "start_frame": 0, "end_frame": 5}]))
    monkeypatch.setenv("VPD_DIVING48_TEST_FILE", str(dv))
    assert get_test_prefixes("diving48") == ("a", "b")
    monkeypatch.setenv("VPD_DIVING48_TEST_FILE", str(tmp_path / "absent.json"))
    with pytest.raises(FileNotFoundError):
        get_test_prefixes("diving48")


def test_pretrained_checkpoint_lookup_prefers_the_v1_file_and_refuses_to_guess(tmp_path, monkeypatch):
    """`--pretrained` (reference models/rgb.py:57-58) reads the file torchvision would download: with several checkpoints of one
    architecture in a directory the ImageNet-V1 file is taken, and a set without it is an error, not a lexicographic pick."""
    import torch
    from vpd_amd.models.rgb import IMAGENET_V1_FILES, find_imagenet_weights
    monkeypatch.setattr(torch.hub, "get_dir", lambda: str(tmp_path / "hub"))
    monkeypatch.setenv("VPD_PRETRAINED_WEIGHTS", str(tmp_path))
    with pytest.raises(FileNotFoundError):
        find_imagenet_weights("resnet50")
    (tmp_path / "resnet50-11ad3fa6.pth").write_bytes(b"v2")
    assert find_imagenet_weights("resnet50").endswith("resnet50-11ad3fa6.pth")       # a single candidate is taken as it is
    (tmp_path / "resnet50-00000000.pth").write_bytes(b"other")
    with pytest.raises(FileNotFoundError, match="none is the ImageNet-V1 file"):
        find_imagenet_weights("resnet50")
    (tmp_path / IMAGENET_V1_FILES["resnet50"]).write_bytes(b"v1")
    assert find_imagenet_weights("resnet50").endswith(IMAGENET_V1_FILES["resnet50"])
    assert not find_imagenet_weights("resnet50").endswith("wide_resnet50_2-95faca4d.pth")
    monkeypatch.setenv("VPD_PRETRAINED_WEIGHTS", str(tmp_path / "resnet50-11ad3fa6.pth"))   # an explicit file always wins
    assert find_imagenet_weights("resnet50").endswith("resnet50-11ad3fa6.pth")
    (tmp_path / "hub" / "checkpoints").mkdir(parents=True)
    (tmp_path / "hub" / "checkpoints" / "resnet18-f37072fd.pth").write_bytes(b"v1")
    monkeypatch.delenv("VPD_PRETRAINED_WEIGHTS")
    assert find_imagenet_weights("resnet18").endswith("resnet18-f37072fd.pth")


def test_streaming_writer_merges_late_frames_of_a_flushed_video(tmp_path):
    """StreamingWriter (row f4; reference apply_vpd_model.py:171-178 writes every video once, at the end): a video whose frame
    count was under-estimated is flushed early; frames that arrive afterwards must end up in the SAME pickle, sorted, not in a
    file that holds the late frames only."""
    from vpd_amd.apply import StreamingWriter
    w = StreamingWriter(str(tmp_path), ["a", "b"], [2, 3])          # "a" really has 4 frames
    e = lambda f: (f, np.full((2, 4), f, np.float32), {})
    w.add_many(0, [e(0), e(1)])                                     # -> flushed (count reached)
    w.add_many(1, [e(0), e(1)])
    w.add_many(0, [e(3), e(2)])                                     # late frames: a second flush of "a"
    w.add(1, e(2))
    w.close()
    assert sorted(w.written) == ["a", "b"]
    with open(os.path.join(str(tmp_path), "a.emb.pkl"), "rb") as fp:
        got = pickle.load(fp)
    assert [g[0] for g in got] == [0, 1, 2, 3] and all(float(g[1][0, 0]) == g[0] for g in got)
    with open(os.path.join(str(tmp_path), "b.emb.pkl"), "rb") as fp:
        assert [g[0] for g in pickle.load(fp)] == [0, 1, 2]
