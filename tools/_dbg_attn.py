import sys, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from dino_amd import capi
import test_ops_gpu as T
lib = capi.lib()
for planes in (1, 2):
    for case in ((2, 2, 197, 5, False), (1, 1, 300, 77, True)):
        B, H, ntok, seed, spike = case
        outs = {}
        for v in (0, 0, 1, 3):
            capi.check(lib.dinoseg_set_option(b"attn_variant", v))
            got, ref, lse, _ = T._attention_case(B, H, ntok, planes, seed, spike)
            outs.setdefault(v, []).append((got, lse))
        a, b = outs[0]
        print(planes, case, "base-base", float((a[0] - b[0]).abs().max()), float((a[1] - b[1]).abs().max()))
        for v in (1, 3):
            d = (outs[v][0][0] - a[0]).abs()
            print(planes, case, "var", v, float(d.max()), float((outs[v][0][1] - a[1]).abs().max()), "rows differing", int((d.amax(1) > 0).sum()), "of", d.shape[0],
                  "first", (d.amax(1) > 0).nonzero()[:5].flatten().tolist(), "cols", (d.amax(0) > 0).nonzero()[:8].flatten().tolist())
