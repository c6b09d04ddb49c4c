#!/usr/bin/env python3
"""Measured dense-bf16 MFMA rate of this chip (a measured peak next to the vendor's 2.5 PFLOP/s).

A register-only v_mfma_f32_32x32x16_bf16 loop (csrc/mfma_peak.hip) on 1..8 resident waves per SIMD, with 1 / 4 independent
accumulator chains per wave, on all-zero and on pseudo-random operands.   python tools/mfma_peak.py  (GPU box)
Under rocprofv3 --pmc GRBM_GUI_ACTIVE (tools/pmc_ops.sh-style) the same launches give the clock each case runs at."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from dino_amd import capi  # noqa: E402

lib = C.CDLL(os.path.join(os.path.dirname(capi.LIB_PATH), "libdinoseg_tools.so"))      # make -C dino_amd/csrc tools
lib.dinoseg_tools_mfma_peak.restype = C.c_int
lib.dinoseg_tools_mfma_peak.argtypes = [C.c_int32, C.c_int32, C.c_uint32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
scratch = torch.zeros(16, device="cuda")
flops = C.c_double(0.0)
ITERS = int(os.environ.get("MFMA_ITERS", "20000"))


def run(wps, seed, chains):
    def f():
        rc = lib.dinoseg_tools_mfma_peak(wps, ITERS, seed, chains, scratch.data_ptr(), C.addressof(flops), capi.stream_ptr())
        assert rc == 0, rc
    for _ in range(2):
        f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(3):
        f()
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 3
    return flops.value / (ms * 1e-3) / 1e12, ms


print(f"{'waves/SIMD':>10} {'mode':>22} {'operands':>9} {'TFLOP/s':>9} {'of 2500':>8} {'ms':>8}")
ONLY = os.environ.get("MFMA_MODES")
MODES = [(4, "4 chains"), (1, "1 chain")] + [(100 + n, f"+{n} VALU / 4 MFMA") for n in (8, 16, 32)] + \
        [(200 + n, f"+{n} SALU / 4 MFMA") for n in (8, 16, 32)] + [(300 + n, f"+{n} s_nop / 4 MFMA") for n in (16, 32)] + \
        [(400 + n, f"+{n} s_waitcnt / 4 MFMA") for n in (16, 32)] + \
        [(800, "gap: bare MFMA"), (801, "gap: +2 exp"), (802, "gap: +2 exp 2 add"), (803, "gap: +2exp 2add 1cvt"), (804, "tile: not interleaved")] + \
        [(500, "32x32x16 regs (ctl)"), (504, "16x16x32 regs"), (604, "32x32x16 LDS-fed"), (704, "16x16x32 LDS-fed")]
if ONLY:
    MODES = [m for m in MODES if str(m[0]) in ONLY.split(",")]
for chains, label in MODES:
    for wps in (1, 2, 3, 4, 8):
        for seed, name in ((0, "zero"), (7, "random")) if (chains < 100 or chains >= 500) else ((0, "zero"),):
            tf, ms = run(wps, seed, chains)
            print(f"{wps:>10} {label:>22} {name:>9} {tf:9.1f} {tf / 2500:8.3f} {ms:8.2f}", flush=True)
