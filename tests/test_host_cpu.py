"""CPU tests of the host side: C-ABI library loads and exports every declared symbol, the DINOSeg mirror
keeps the reference's surface (state_dict keys, ValueError text, checkpoint round trip) and fails loudly
without a GPU (no CPU fallback)."""
import ctypes
import os

import numpy as np
import pytest
import torch

import dino_amd
from dino_amd import DINOSeg, ViTConfig, capi, procedural_state_dict
from dino_amd.ckpt import read_checkpoint, save_checkpoint
from dino_amd.weights import synthetic_frames, tensor_shapes


def test_library_exports_every_header_symbol():
    lib = capi.lib()
    declared = capi.header_symbols()
    assert len(declared) >= 17
    assert set(declared) == set(capi.SIGNATURES), "capi.SIGNATURES must mirror include/dinoseg.h"
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.dinoseg_version() >= 100


def test_handle_lifecycle_and_errors_without_gpu():
    lib = capi.lib()
    h = ctypes.c_void_p()
    bad = capi.Config(100, 2, 1, 8, 4, 7, capi.HEAD_MLP, 28, 1e-6, capi.BF16)
    assert lib.dinoseg_create(ctypes.byref(bad), ctypes.byref(h)) == -1
    assert "unsupported config" in capi.last_error()
    cfg = capi.Config(384, 6, 1, 8, 4, 7, capi.HEAD_MLP, 28, 1e-6, capi.BF16X3)
    assert lib.dinoseg_create(ctypes.byref(cfg), ctypes.byref(h)) == 0
    assert lib.dinoseg_workspace_bytes(h, 1, 480) > 3601 * 384 * 4
    assert lib.dinoseg_workspace_bytes(h, 1, 250) == -1
    assert lib.dinoseg_prepare_resolution(h, 250, None) == -1
    with pytest.raises(ValueError, match="Resolution should be a multiple of 8."):
        capi.check(-1)
    # strict binding: unknown key and wrong shape are rejected
    buf = (ctypes.c_float * 4)()
    shape = (ctypes.c_int64 * 1)(4)
    assert lib.dinoseg_bind_weight(h, b"dino.nope", ctypes.addressof(buf), shape, 1) == -1
    assert lib.dinoseg_bind_weight(h, b"dino.norm.weight", ctypes.addressof(buf), shape, 1) == -1
    assert lib.dinoseg_refresh_weights(h, None) == -3          # missing keys
    assert "missing key" in capi.last_error()
    assert lib.dinoseg_state_generation(h) == 0                # nothing allocated yet
    assert lib.dinoseg_destroy(h) == 0
    # the four precisions (include/dinoseg.h) create a handle; anything else is refused; the process-wide switches validate their values
    for prec in (capi.BF16, capi.BF16X3, capi.FP16, capi.FP16X3):
        cfg = capi.Config(384, 6, 1, 8, 4, 7, capi.HEAD_MLP, 28, 1e-6, prec)
        assert lib.dinoseg_create(ctypes.byref(cfg), ctypes.byref(h)) == 0 and lib.dinoseg_destroy(h) == 0
    cfg = capi.Config(384, 6, 1, 8, 4, 7, capi.HEAD_MLP, 28, 1e-6, 4)
    assert lib.dinoseg_create(ctypes.byref(cfg), ctypes.byref(h)) == -1
    assert lib.dinoseg_set_option(b"op_fmt", 2) == -1 and "op_fmt" in capi.last_error()
    assert lib.dinoseg_set_option(b"op_fmt", 0) == 0 and lib.dinoseg_set_option(b"mlp_variant", 1) == 0
    assert lib.dinoseg_set_option(b"no_such_option", 1) == -1
    assert set(dino_amd.dinoseg._PRECISIONS) == {"bf16", "bf16x3", "fp16", "fp16x3"}
    with pytest.raises(ValueError, match="precision"):
        DINOSeg(precision="fp32")


def test_state_dict_keys_match_reference_schema():
    for head, L in (("mlp", 3), ("linear", 1)):
        m = DINOSeg(head=head, n_blocks=L)
        want = tensor_shapes(ViTConfig(n_blocks=L, head=head))
        got = {k: tuple(v.shape) for k, v in m.state_dict().items()}
        assert got == dict(want)
    assert len(DINOSeg(head="mlp", n_blocks=3).state_dict()) == 48      # 48 tensors for L=3
    assert sum(v.numel() for v in DINOSeg(head="mlp", n_blocks=3).state_dict().values()) == 5797903


def test_set_resolution_and_no_cpu_fallback():
    m = DINOSeg(head="mlp", n_blocks=1)
    with pytest.raises(ValueError, match="Resolution should be a multiple of 8."):
        m.set_resolution(250)
    m.set_resolution(240)
    assert m.resolution == 240 and m.transforms.resolution == 240
    assert m.device.type == "cpu"
    with pytest.raises(capi.DinosegError, match="no CPU path"):
        m(torch.zeros(1, 3, 64, 64))
    with pytest.raises(capi.DinosegError, match="no CPU path"):
        m.predict(np.zeros((240, 240, 3), np.uint8))


def test_transforms_match_oracle_preprocess():
    from oracle import dinoseg_oracle as O
    fr = synthetic_frames(1, 64, seed=2)
    t = dino_amd.get_transforms(64)(image=fr[0])["image"]
    assert t.dtype == torch.float32 and tuple(t.shape) == (3, 64, 64)
    assert torch.equal(t, O.preprocess(fr)[0])
    big = dino_amd.get_transforms(32).resize(fr[0])
    assert big.shape == (32, 32, 3) and big.dtype == np.uint8


def test_checkpoint_round_trip(tmp_path):
    sd = procedural_state_dict(ViTConfig(n_blocks=2))
    m = DINOSeg(data_path="d", write_path="w", head="mlp", n_blocks=2, optimizer=torch.optim.Adam, lr=1e-3)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    path = os.path.join(tmp_path, "two_block.ckpt")
    save_checkpoint(m, path, epoch=3)
    ck = read_checkpoint(path)
    assert ck["pytorch-lightning_version"] == "1.5.10" and ck["hyper_parameters"]["n_blocks"] == 2
    m2 = DINOSeg.load_from_checkpoint(path)
    assert m2.n_blocks == 2 and m2.head == "mlp" and m2.optimizer is torch.optim.Adam and m2.lr == 1e-3
    for k, v in m2.state_dict().items():
        assert np.array_equal(v.numpy(), sd[k]), k
    with pytest.raises(RuntimeError):        # strict load, as the reference
        m2.load_state_dict({"dino.cls_token": torch.zeros(1, 1, 384)}, strict=True)


def test_checkpoint_with_unimportable_hyperparameters(tmp_path):
    """PL checkpoints pickle objects of packages that are absent here (comet logger...): they must not block loading."""
    import pickle
    import types
    fake = types.ModuleType("comet_like_pkg")

    class Logger:
        def __init__(self):
            self.key = "secret"
    Logger.__module__ = "comet_like_pkg"
    Logger.__qualname__ = "Logger"
    fake.Logger = Logger
    import sys
    sys.modules["comet_like_pkg"] = fake
    sd = procedural_state_dict(ViTConfig(n_blocks=1))
    path = os.path.join(tmp_path, "pl.ckpt")
    torch.save({"state_dict": {k: torch.from_numpy(v) for k, v in sd.items()},
                "hyper_parameters": {"head": "mlp", "n_blocks": 1, "comet_logger": Logger(), "data_path": "x",
                                     "write_path": "y", "optimizer": torch.optim.AdamW}}, path, pickle_module=pickle)
    del sys.modules["comet_like_pkg"]
    m = DINOSeg.load_from_checkpoint(path)
    assert m.n_blocks == 1 and m.comet_logger is None and m.optimizer is torch.optim.AdamW


def test_resize_mirror_matches_scalar_restatement():
    """dino_amd.preprocess (host mirror of cv2.INTER_LINEAR on uint8) against the pixel-at-a-time restatement in oracle/:
    upscale, downscale, the exact-2x INTER_AREA switch, 1-pixel sources and a known 2x ramp (taps 0.25 / 0.75)."""
    from dino_amd.preprocess import resize_linear_u8
    from oracle.resize_oracle import resize_linear_u8 as ref
    rng = np.random.default_rng(0)
    for sh, sw, dh, dw in [(5, 7, 8, 8), (12, 16, 8, 8), (16, 16, 8, 8), (3, 3, 16, 16), (1, 1, 4, 4), (48, 64, 40, 40), (9, 4, 16, 24)]:
        img = rng.integers(0, 256, (sh, sw, 3), dtype=np.uint8)
        assert np.array_equal(resize_linear_u8(img, dh, dw), ref(img, dh, dw)), (sh, sw, dh, dw)
    ramp = np.tile(np.array([0, 16, 32, 48], dtype=np.uint8)[None, :, None], (2, 1, 3))
    assert resize_linear_u8(ramp, 2, 8)[0, :, 0].tolist() == [0, 4, 12, 20, 28, 36, 44, 48]
    const = np.full((10, 13, 3), 200, np.uint8)
    assert np.unique(resize_linear_u8(const, 16, 16)).tolist() == [200]
    same = rng.integers(0, 256, (8, 8, 3), dtype=np.uint8)
    assert resize_linear_u8(same, 8, 8) is not None and np.array_equal(resize_linear_u8(same, 8, 8), same)


def test_fast_signature_sees_every_kind_of_weight_change():
    """The cheap per-call check of DINOSeg._sync_weights (no state_dict walk) must notice everything the full signature does:
    in-place updates, optimizer steps, load_state_dict, replaced Parameters / sub-modules, added entries, invalidate_weights()."""
    import copy
    import pickle
    m = DINOSeg(head="mlp", n_blocks=2)
    assert m._fast_signature() is None                       # nothing indexed before the first bind
    m._fast_index = m._build_fast_index()
    base = m._fast_signature()
    assert base is not None and m._fast_signature() == base  # stable while nothing changes
    with torch.no_grad():
        m.clf.layer_3.bias.add_(1.0)
    assert m._fast_signature() != base
    base = m._fast_signature()
    opt = torch.optim.SGD(m.parameters(), lr=0.1)
    for p in m.parameters():
        p.grad = torch.zeros_like(p)
    opt.step()
    assert m._fast_signature() != base
    base = m._fast_signature()
    m.load_state_dict({k: torch.from_numpy(v) for k, v in procedural_state_dict(m.cfg).items()})
    assert m._fast_signature() != base
    base = m._fast_signature()
    m.dino.norm.bias.data = torch.ones_like(m.dino.norm.bias)          # new storage, same Parameter, version unchanged
    assert m._fast_signature() != base
    m._fast_index = m._build_fast_index()
    base = m._fast_signature()
    m.clf.layer_1.bias = torch.nn.Parameter(torch.zeros(200))          # replaced Parameter object
    assert m._fast_signature() is None
    m._fast_index = m._build_fast_index()
    m.dino.blocks[1] = copy.deepcopy(m.dino.blocks[0])                 # replaced sub-module
    assert m._fast_signature() is None
    m._fast_index = m._build_fast_index()
    m.clf.register_buffer("extra", torch.zeros(1))                     # added entry
    assert m._fast_signature() is None
    m._fast_index = m._build_fast_index()
    base = m._fast_signature()
    m.invalidate_weights()
    assert m._fast_signature() != base
    m2 = pickle.loads(pickle.dumps(m))                                  # the index is process state: not pickled
    assert "_fast_index" not in m2.__dict__ and m2._fast_signature() is None


def test_resize_restatement_against_an_independent_bilinear():
    """cv2 is not in this image (oracle/resize_oracle.py: parity unpinned), so the restatement is held against an INDEPENDENT
    implementation of the same sampling rule: torch's float64 bilinear (align_corners=False = cv2's (dx + 0.5) * scale - 0.5 with edge
    clamping, no antialiasing; 'area' for the exact-2x case, where cv2 switches to INTER_AREA).  cv2's uint8 path rounds its taps to
    11 bits and truncates intermediate sums, so the two may differ by one grey level, never by more."""
    import torch.nn.functional as F
    from dino_amd.preprocess import resize_linear_u8
    rng = np.random.default_rng(1)
    for sh, sw, dh, dw in [(5, 7, 8, 8), (12, 16, 8, 8), (3, 3, 16, 16), (48, 64, 40, 40), (9, 4, 16, 24), (300, 400, 240, 240),
                           (480, 640, 480, 480), (16, 16, 8, 8), (96, 128, 48, 64)]:
        img = rng.integers(0, 256, (sh, sw, 3), dtype=np.uint8)
        x = torch.from_numpy(img).permute(2, 0, 1)[None].double()
        if sh == 2 * dh and sw == 2 * dw:
            ref = F.interpolate(x, size=(dh, dw), mode="area")
        else:
            ref = F.interpolate(x, size=(dh, dw), mode="bilinear", align_corners=False, antialias=False)
        ref = ref[0].permute(1, 2, 0).numpy()
        got = resize_linear_u8(img, dh, dw).astype(np.float64)
        d = got - ref
        assert np.abs(d).max() <= 1.0 + 1e-9, (sh, sw, dh, dw, np.abs(d).max())
        # (rounding to a grey level: mean |d| about 0.25; cv2's vertical pass truncates twice -- ((b0 * (S0 >> 4)) >> 16) + ... -- which
        #  shows as a bias of up to about +-0.1 grey levels against exact arithmetic, by the tap values)
        assert abs(d.mean()) < 0.2 and np.abs(d).mean() < 0.3, (sh, sw, dh, dw, d.mean(), np.abs(d).mean())


def test_host_side_under_address_sanitizer():
    """SURVEY.md section 5 (sanitizers): the host side of the C-ABI library built with -fsanitize=address (`make -C dino_amd/csrc asan`:
    device code objects as usual, every host function instrumented) runs this file's C-ABI tests -- symbol table, handle lifecycle,
    refused arguments, strict binding, option validation -- in a child process with the ASan runtime preloaded.  Build container only
    (needs hipcc; the GPU box never runs a sanitizer build)."""
    import shutil
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    csrc = os.path.join(root, "dino_amd", "csrc")
    if shutil.which("make") is None or not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("no toolchain")
    subprocess.run(["make", "-C", csrc, "-j", str(min(8, os.cpu_count() or 1)), "asan"], check=True, capture_output=True)
    rt = subprocess.run(["make", "-s", "-C", csrc, "asan-runtime"], check=True, capture_output=True, text=True).stdout.strip()
    lib = os.path.join(root, "dino_amd", "lib", "libdinoseg_hip_asan.so")
    assert os.path.exists(rt) and os.path.exists(lib)
    syms = subprocess.run(["nm", "-D", lib], capture_output=True, text=True).stdout
    assert "__asan_init" in syms, "the library is not instrumented"
    env = dict(os.environ, LD_PRELOAD=rt, DINOSEG_LIB=lib, ASAN_OPTIONS="detect_leaks=0:halt_on_error=1")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-x", "-q", "-p", "no:cacheprovider",
                        "-k", "exports or lifecycle or no_cpu_fallback"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "AddressSanitizer" not in (r.stdout + r.stderr), (r.stdout + r.stderr)[-2000:]
    assert "3 passed" in r.stdout


def test_generated_attention_bodies_are_current():
    """dino_amd/csrc/attention_za_gen.inc is what tools/gen_attn_asm.py writes (the generator re-derives every body -- register map,
    pipeline order, counted LDS waits, with its own consistency assertions at every label -- and compares with the committed file)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "gen_attn_asm.py"), "--check"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
