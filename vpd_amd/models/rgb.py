"""RGBF_EmbeddingModel: the reference's student wrapper (models/rgb.py:46-86)
over the HIP engine.  Same constructor arguments, attributes, state_dict keys,
embed() contract and assertion behaviour; the arithmetic is libvpdhip."""
import math

import numpy as np
import torch
import torch.nn as nn

from ..engine import StudentEngine
from .module import ENCODER_ARCH, attach_views


# The ImageNet-V1 files `pretrained=True` downloads (torchvision's model_urls; models/rgb.py:57-58 reaches them through
# ENCODER_ARCH[...].pretrained_init): preferred when a directory holds several checkpoints of one architecture.
IMAGENET_V1_FILES = {
    "resnet18": "resnet18-f37072fd.pth", "resnet34": "resnet34-b627a593.pth", "resnet50": "resnet50-0676ba61.pth",
    "resnet101": "resnet101-63fe2227.pth", "resnet152": "resnet152-394f9c45.pth",
    "wide_resnet50_2": "wide_resnet50_2-95faca4d.pth", "wide_resnet101_2": "wide_resnet101_2-32ee1156.pth",
}


def find_imagenet_weights(model_arch):
    """The checkpoint torchvision would download for `model_arch`: $VPD_PRETRAINED_WEIGHTS (a file, or a directory holding
    <arch>*.pth), else torch hub's cache (~/.cache/torch/hub/checkpoints/<arch>-<hash>.pth).  There is no download here.
    With several candidates in one place the ImageNet-V1 file (what the reference's pretrained=True means) is taken; if
    none of them is that file the choice would be a guess, so it is an error."""
    import glob
    import os
    env = os.environ.get("VPD_PRETRAINED_WEIGHTS")
    if env and os.path.isfile(env):
        return env
    places = ([env] if env else []) + [os.path.join(torch.hub.get_dir(), "checkpoints")]
    for place in places:
        cand = sorted(glob.glob(os.path.join(place, model_arch + "-*.pth")) + glob.glob(os.path.join(place, model_arch + ".pth")))
        if not cand:
            continue
        v1 = [c for c in cand if os.path.basename(c) == IMAGENET_V1_FILES.get(model_arch)]
        if len(cand) > 1 and not v1:
            raise FileNotFoundError("--pretrained: %d checkpoints for %s in %s and none is the ImageNet-V1 file %s; point "
                                    "VPD_PRETRAINED_WEIGHTS at the file to use" % (len(cand), model_arch, place,
                                                                                   IMAGENET_V1_FILES.get(model_arch)))
        path = (v1 or cand)[0]
        print("--pretrained: ImageNet weights from", path)
        return path
    raise FileNotFoundError("--pretrained: no ImageNet state_dict for %s; set VPD_PRETRAINED_WEIGHTS to the torchvision "
                            "checkpoint (or to a directory holding %s-*.pth)" % (model_arch, model_arch))


class RGBF_EmbeddingModel(nn.Module):
    """Basic embedding model with single frame features (HIP / MI355X)."""

    def __init__(self, model_arch, emb_dim, use_flow, device, pretrained=False, in_channels=None, dtype="bf16"):
        """Reference signature (models/rgb.py:49-50) + in_channels: an explicit input-channel count (1..8) for the
        variants the reference's hard-coded 5 cannot express -- BASELINE configs[2] runs a 6-channel two-stream input.
        The stem is then initialised by the same recipe as add_flow_to_model (models/rgb.py:19-23): the channel mean of
        a 3-channel kaiming kernel expanded to in_channels.
        dtype: "bf16" (default; training and inference) or "fp16" -- the reference's own GPU precision (fp16 autocast,
        train_vpd_model.py:79) for INFERENCE: embed() / the apply loop run on libvpdhip_f16.so, train mode raises."""
        super().__init__()
        if "effnet" in model_arch:
            raise NotImplementedError("EfficientNet students are out of scope (SURVEY.md 2.1 #3)")
        if model_arch not in ENCODER_ARCH:
            raise KeyError(model_arch)
        # pretrained: the reference asks torchvision for ImageNet weights (models/rgb.py:57-58, a download); here the SAME
        # checkpoint file is read from disk -- VPD_PRETRAINED_WEIGHTS, else torchvision's hub cache -- see load_imagenet_backbone
        weights = find_imagenet_weights(model_arch) if pretrained else None
        self.device = device
        self.use_flow = use_flow
        self.emb_dim = emb_dim
        self.model_arch = model_arch
        c_in = 5 if use_flow else 3
        if in_channels is not None:
            c_in = int(in_channels)
            if not 1 <= c_in <= 8:
                raise ValueError("in_channels must be in 1..8")
        self.in_channels = c_in
        eng = StudentEngine(model_arch, c_in, emb_dim, device="cuda" if str(device) == "cuda" else device, dtype=dtype)
        self._engine_ref = [eng]
        attach_views(self, eng.enc_names, eng.view)
        for k in eng.enc_names:
            self.get_parameter(k).grad = eng.view(k, eng.grads)
        # BN buffers in reference order: running_mean, running_var, num_batches_tracked right after weight/bias
        for i, bn in enumerate(eng.bn_names):
            rm, rv = eng.bn_views(bn)
            mod = self.get_submodule(bn)
            mod.register_buffer("running_mean", rm)
            mod.register_buffer("running_var", rv)
            mod.register_buffer("num_batches_tracked", eng.num_batches_tracked[i])
        self.reset_parameters()
        if weights is not None:
            self.load_imagenet_backbone(torch.load(weights, map_location="cpu", weights_only=True))

    def load_imagenet_backbone(self, sd):
        """What `pretrained=True` does in the reference (models/rgb.py:57-61), from a torchvision-format state_dict
        (`conv1.weight`, `layer1.0.bn1.running_mean`, ..., `fc.weight [1000, F]`): every backbone tensor is taken as it
        is, the 3-channel stem becomes its channel mean expanded to the input channels when they are not 3
        (add_flow_to_model, :19-23), and the 1000-way fc is dropped -- the embedding layer keeps its fresh nn.Linear
        initialisation (replace_last_layer, :40-43).  Checkpoints without `num_batches_tracked` (saved before torch 0.4.1,
        as the V1 files are) load like they do in nn.BatchNorm: the counters keep their value."""
        own = super().state_dict()
        new = {}
        for k, v in sd.items():
            if k.startswith("fc."):
                continue
            key = "resnet." + k
            if key not in own:
                raise KeyError("unexpected tensor in the ImageNet state_dict: " + k)
            if k == "conv1.weight" and self.in_channels != v.shape[1]:
                v = v.mean(dim=1, keepdim=True).expand(-1, self.in_channels, -1, -1).contiguous()
            if tuple(v.shape) != tuple(own[key].shape):
                raise ValueError("shape of %s: %s, expected %s" % (k, tuple(v.shape), tuple(own[key].shape)))
            new[key] = v
        # torchvision's ImageNet-V1 files predate BatchNorm's num_batches_tracked buffer; nn.BatchNorm's own loader
        # tolerates its absence (the counter keeps its value), and so does this one
        for k in own:
            if k.endswith(".num_batches_tracked") and k not in new:
                new[k] = own[k]
        missing = [k for k in own if k not in new and not k.startswith("resnet.fc.")]
        if missing:
            raise KeyError("the ImageNet state_dict lacks " + ", ".join(missing[:4]))
        new["resnet.fc.weight"], new["resnet.fc.bias"] = own["resnet.fc.weight"], own["resnet.fc.bias"]
        self.load_state_dict(new)
        self.engine.mark_weights_changed()

    @property
    def engine(self):
        return self._engine_ref[0]

    def reset_parameters(self, seed=None):
        """Reference init: kaiming-normal fan_out convs, BN gamma=1/beta=0 (models/module.py:71-76);
        5-ch stem = channel mean of a 3-ch kernel (models/rgb.py:19-23); nn.Linear default fc."""
        eng = self.engine
        g = None
        if seed is not None:
            g = torch.Generator(device=eng.device).manual_seed(seed)
        with torch.no_grad():
            for name in eng.enc_names:
                p = self.get_parameter(name)
                kind = eng.layout[name][0]
                if kind == 0:
                    co, ci, kh, kw = p.shape
                    std = math.sqrt(2.0 / (co * kh * kw))
                    if name == "resnet.conv1.weight" and self.in_channels != 3:
                        w3 = torch.randn((co, 3, kh, kw), device=eng.device, generator=g) * std
                        p.copy_(w3.mean(dim=1, keepdim=True).expand_as(p))
                    else:
                        p.copy_(torch.randn(p.shape, device=eng.device, generator=g) * std)
                elif kind == 1:
                    p.fill_(1.0)
                elif kind == 2:
                    p.zero_()
                else:
                    bound = 1.0 / math.sqrt(self.get_parameter("resnet.fc.weight").shape[1])
                    p.copy_((torch.rand(p.shape, device=eng.device, generator=g) * 2 - 1) * bound)
            for bn in eng.bn_names:
                rm, rv = eng.bn_views(bn)
                rm.zero_()
                rv.fill_(1.0)
            eng.num_batches_tracked.zero_()      # (the property: pending host-side counts are folded in first, then cleared)

    def state_dict(self, *args, **kwargs):
        # the BatchNorm modules' num_batches_tracked buffers are views of the engine's counter tensor, which is brought up
        # to date when read through the engine (StudentEngine.num_batches_tracked)
        _ = self.engine.num_batches_tracked
        return super().state_dict(*args, **kwargs)

    def load_state_dict(self, *args, **kwargs):
        _ = self.engine.num_batches_tracked      # fold pending counts in before the buffers are overwritten
        return super().load_state_dict(*args, **kwargs)

    def _check_storage(self):
        p = self.get_parameter("resnet.conv1.weight")
        if p.data_ptr() != self.engine.params.data_ptr():
            raise RuntimeError("model parameters were moved off the engine's flat buffer (.cpu()/.to(other)); "
                               "the HIP student must stay on its GPU")

    def forward(self, x):
        """f32 [N,C,H,W] on the GPU -> f32 [N,emb_dim].  Train mode uses batch statistics and
        updates the running ones (nn.BatchNorm2d semantics); gradients flow only through
        ModelTrainer.epoch's fused step, not through torch autograd."""
        self._check_storage()
        x = x.to(self.engine.device, dtype=torch.float32).contiguous()
        if self.training:
            return self.engine.forward_train(x, None, motion=False)
        return self.engine.forward_eval(x)

    def embed(self, x):
        if not isinstance(x, torch.Tensor):
            x = torch.as_tensor(np.asarray(x), dtype=torch.float32)
        x = x.to(self.engine.device)
        if len(x.shape) == 3:
            x = x.unsqueeze(0)

        if self.in_channels not in (3, 5):
            assert x.shape[1] == self.in_channels, 'Wrong number of channels'
        elif self.use_flow:
            assert x.shape[1] == 5, 'Wrong number of channels for RGB + flow'
        else:
            assert x.shape[1] == 3, 'Wrong number of channels for RGB'

        self.eval()
        with torch.no_grad():
            return self(x).cpu().numpy()
