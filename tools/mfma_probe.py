"""MFMA ceiling of the part, measured: register-only v_mfma_f32_32x32x16_bf16 loops (mfma_probe.hip) at several grid
sizes and waves per SIMD.  Build first (on the CPU box): python mfma_probe.py build"""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(ROOT, "mfma_probe.so")
if len(sys.argv) > 1 and sys.argv[1] == "build":
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC",
                           os.path.join(ROOT, "mfma_probe.hip"), "-o", SO])
    sys.exit(0)
import torch
lib = ctypes.CDLL(SO)
lib.mfma_probe.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
lib.mfma_probe16.argtypes = lib.mfma_probe.argtypes
out = torch.zeros(4, device="cuda")
iters = 20000
for threads in (256, 512):
    for wgs in (32, 64, 128, 256, 512):
        def run():
            assert lib.mfma_probe(wgs, threads, iters, out.data_ptr(), torch.cuda.current_stream().cuda_stream) == 0
        run(); torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); run(); run(); e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 2
        waves = wgs * threads // 64
        fl = waves * iters * 8 * 2.0 * 32 * 32 * 16
        per_mfma_ns = ms * 1e6 / (iters * 8) / (threads // 256)      # per MFMA per SIMD (threads/256 waves share a SIMD)
        def run16():        # 16 x v_mfma_f32_16x16x32_bf16 per round = the same FLOPs as 8 x 32x32x16
            assert lib.mfma_probe16(wgs, threads, iters, out.data_ptr(), torch.cuda.current_stream().cuda_stream) == 0
        run16(); torch.cuda.synchronize()
        e0.record(); run16(); run16(); e1.record(); torch.cuda.synchronize()
        ms16 = e0.elapsed_time(e1) / 2
        print(f"{wgs:4d} workgroups x {threads // 64} waves: {ms:8.3f} ms  {fl / ms / 1e9:7.0f} TFLOP/s  "
              f"{per_mfma_ns:6.2f} ns per MFMA per SIMD = {32 / per_mfma_ns:.2f} GHz-equivalent at 32 cycles | "
              f"16x16x32: {ms16:8.3f} ms {fl / ms16 / 1e9:7.0f} TFLOP/s", flush=True)
