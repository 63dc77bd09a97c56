"""Does a hipMemsetAsync captured into a HIP graph run, in order and with its value, on every replay?

memset(buf, 0) -> buf += 1 -> out = buf, captured once and replayed several times: `out` must read 1 every time.  A memset
node that is skipped, reordered or replayed with another value shows as a wrong NUMBER, never as a wrong address.  Sizes
and alignments include those of tfc_csr_build (a 4-byte memset at a 4-byte offset, an (N + 1) x 4-byte one).  Variants:
the graph as one chain (one-stream capture) or with a forked side branch (as the default two-stream step has), and
torch's own zero_() in place of the runtime call.  (Lead on the known issue of DESIGN.md section 4.)"""
import ctypes as C
import sys

import torch

hip = C.CDLL("libamdhip64.so")
hip.hipMemsetAsync.argtypes = [C.c_void_p, C.c_int, C.c_size_t, C.c_void_p]
hip.hipMemsetAsync.restype = C.c_int

dev = torch.device("cuda:0")
torch.zeros(1, device=dev)
side = torch.cuda.Stream()
bad = 0
for fork in (False, True):
    for how in ("hipMemsetAsync", "zero_"):
        for n_words, off in ((1, 1), (2, 0), (64, 0), (1024, 0), (2049, 0), (4096, 0), (6913, 0), (65536, 0), (1 << 22, 0)):
            buf = torch.full((n_words + off + 3,), 7, dtype=torch.int32, device=dev)
            out = torch.zeros(n_words, dtype=torch.int32, device=dev)
            junk = torch.zeros(1 << 16, device=dev)
            view = buf[off : off + n_words]

            def body():
                cur = torch.cuda.current_stream()
                if fork:  # an independent side branch, joined at the end
                    side.wait_stream(cur)
                    with torch.cuda.stream(side):
                        junk.add_(1.0)
                if how == "zero_":
                    view.zero_()
                else:
                    rc = hip.hipMemsetAsync(C.c_void_p(view.data_ptr()), 0, 4 * n_words, C.c_void_p(cur.cuda_stream))
                    assert rc == 0, rc
                view.add_(1)
                out.copy_(view)
                if fork:
                    cur.wait_stream(side)

            s = torch.cuda.Stream()
            with torch.cuda.stream(s):
                body()
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                body()
            seen = []
            for rep in range(4):
                g.replay()
                torch.cuda.synchronize()
                seen.append((int(out.min()), int(out.max()), int(buf[off + n_words])))
            ok = all(a == 1 and b == 1 and c == 7 for a, b, c in seen)
            bad += not ok
            print(f"fork {int(fork)} {how:15s} words {n_words:8d} offset {off}: {'ok' if ok else 'WRONG ' + str(seen)}")
print("wrong cases:", bad)
sys.exit(0)
