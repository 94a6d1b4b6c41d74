"""What it costs to get N = 1024 observations (115.6 MB of f32) into host memory: torch's pinned copy, hipMemcpyAsync D2H called
directly, and the fovea kernel storing STRAIGHT into mapped pinned host memory (the C ABI takes any device-visible pointer)."""
import ctypes as C, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "active-gym_amd"), REPO]
import torch, bench
dev = torch.device("cuda:0")
n = 1024
hip = C.CDLL("libamdhip64.so")
pipe = bench.make_pipeline("fixed", n, dev)
frames, cmds, acts = bench.synth_inputs(torch, dev, n, 8, 1234)
obs = torch.empty(pipe.obs_shape, dtype=torch.float32, device=dev)
loc = torch.empty((n, 2), dtype=torch.int32, device=dev)
nbytes = obs.numel() * 4
hs = [torch.empty(pipe.obs_shape, dtype=torch.float32).pin_memory() for _ in range(2)]
for k in range(20):
    pipe.ingest(frames[k % 8], cmds[k % 8]); pipe.fovea(acts[k % 8], out=obs, loc_out=loc)
torch.cuda.synchronize()


def timed(fn, reps=10):
    fn(0); torch.cuda.synchronize()
    t = time.perf_counter()
    for k in range(reps):
        fn(k)
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps


t = timed(lambda k: hs[k % 2].copy_(obs, non_blocking=True))
print("torch pinned copy_ (non_blocking), back to back: %.2f ms = %.1f GB/s" % (t * 1e3, nbytes / t / 1e9))
st = torch.cuda.current_stream().cuda_stream
hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
t = timed(lambda k: hip.hipMemcpyAsync(hs[k % 2].data_ptr(), obs.data_ptr(), nbytes, 2, st))
print("hipMemcpyAsync D2H, back to back: %.2f ms = %.1f GB/s" % (t * 1e3, nbytes / t / 1e9))
# H2D for comparison
hf = torch.empty(frames[0].shape, dtype=torch.uint8).pin_memory()
t = timed(lambda k: frames[0].copy_(hf, non_blocking=True))
print("torch pinned H2D of one step's whole screens (206 MB): %.2f ms = %.1f GB/s" % (t * 1e3, hf.numel() / t / 1e9))
# the kernel writes into host memory itself
dp = C.c_void_p()
rc = hip.hipHostGetDevicePointer(C.byref(dp), C.c_void_p(hs[0].data_ptr()), 0)
print("hipHostGetDevicePointer rc", rc, "device pointer == host pointer:", dp.value == hs[0].data_ptr())
dps = []
for h in hs:
    q = C.c_void_p(); hip.hipHostGetDevicePointer(C.byref(q), C.c_void_p(h.data_ptr()), 0); dps.append(q)
from active_gym import _native as nat
lib = pipe._lib


def zc(k):
    a = acts[k % 8]
    nat.check(lib.agx_fovea_fixed(pipe._ctx, C.c_void_p(a.data_ptr()), nat.DT_F32, None, dps[k % 2], C.c_void_p(loc.data_ptr()), C.c_void_p(st)), pipe._ctx)


t = timed(zc)
print("k_fovea_fixed storing straight into mapped pinned host memory: %.2f ms = %.1f GB/s" % (t * 1e3, nbytes / t / 1e9))
pipe.fovea(acts[9 % 8], out=obs, loc_out=loc); torch.cuda.synchronize()
zc(9); torch.cuda.synchronize()
print("zero-copy result == device result:", bool(torch.equal(hs[1], obs.cpu())))


def step_copy(k):
    pipe.ingest(frames[k % 8], cmds[k % 8]); pipe.fovea(acts[k % 8], out=obs, loc_out=loc); hs[k % 2].copy_(obs, non_blocking=True)


def step_zc(k):
    pipe.ingest(frames[k % 8], cmds[k % 8]); zc(k)


print("step (device-resident inputs) + pinned copy: %.2f ms; with zero-copy fovea stores: %.2f ms" % (timed(step_copy) * 1e3, timed(step_zc) * 1e3))
