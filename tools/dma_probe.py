"""LDS-DMA throughput per CU and its interference with MFMA issue (tools/dma_probe.hip).  Build first on the CPU box:
python tools/dma_probe.py build.  Each round = 8 x 1 KiB loads per wave (64 KiB per workgroup) + 8 x nm MFMAs per wave."""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(ROOT, "dma_probe.so")
if len(sys.argv) > 1 and sys.argv[1] == "build":
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC",
                           os.path.join(ROOT, "dma_probe.hip"), "-o", SO])
    sys.exit(0)
import torch
lib = ctypes.CDLL(SO)
lib.dma_probe.argtypes = [ctypes.c_int] * 4 + [ctypes.c_void_p] * 3 + [ctypes.c_int, ctypes.c_uint, ctypes.c_int]
CHUNKS = 32768                                                              # x 64 KiB = 2 GiB
src = torch.empty(CHUNKS * 32768, device="cuda", dtype=torch.bfloat16).normal_()
out = torch.zeros(4, device="cuda")
iters = 2000
wgs = 256


def timed(mode, nm, reuse, issuers=8):
    def run():
        assert lib.dma_probe(mode, wgs, iters, nm, src.data_ptr(), out.data_ptr(), torch.cuda.current_stream().cuda_stream,
                             reuse, CHUNKS, issuers) == 0
    run(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); run(); run(); e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 2
    tf = wgs * 8 * iters * 8 * nm * 2.0 * 16 * 16 * 32 / ms / 1e9
    tbs = wgs * iters * 65536 / ms / 1e9 if mode else 0.0
    return f"{ms * 1e3 / iters:6.3f} us/round {tf:5.0f} TF {tbs:5.1f} TB/s"


for nm in (0, 8, 16):
    print(f"nm {nm:2d} | no loads: {timed(0, nm, 0) if nm else '-'}", flush=True)
    for reuse in (0, 8, 5, 2, 1):
        row = [f"{iss} issuers: {timed(1, nm, reuse, iss)}" for iss in (8, 4, 2)]
        print(f"        reuse {reuse} (L2 miss share {'0' if not reuse else f'1/{reuse}'}) | " + " | ".join(row), flush=True)
