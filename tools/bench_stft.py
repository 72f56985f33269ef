"""STFT / ISTFT throughput on the MI355X (device-resident input) next to a plain numpy rfft of the same frames on the host.
Algorithmic bytes per frame: 1024 B in (256 new fp32 samples; 2048 B for float64 audio) + 4104 B out."""
import importlib, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
H = importlib.import_module("disentangled-vae_amd.stft")

def timeit(fn, n=20):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n

def main():
    dev = torch.device("cuda")
    res = {}
    for secs, dt in ((5, torch.float64), (600, torch.float64), (600, torch.float32)):
        n = 16000 * secs
        x = torch.randn(n, dtype=dt, device=dev)
        T = H.frame_count(n, 1024, 256)
        w = H.window_f64("hann", 1024, dev)
        t_c = timeit(lambda: H.stft_device(x, w, 1024, 256, T, 2))           # frame-major complex (the drop-in's stft())
        t_c0 = timeit(lambda: H.stft_device(x, w, 1024, 256, T, 0))          # bin-major complex (stft_pytorch)
        t_p = timeit(lambda: H.stft_device(x, w, 1024, 256, T, 1))
        S = H.stft_device(x, w, 1024, 256, T, 0)
        Sr = H.stft_device(x, w, 1024, 256, T, 2).T
        t_i = timeit(lambda: H.istft_device(Sr, w, 1024, 256, T, 0, n))
        t_i0 = timeit(lambda: H.istft_device(S, w, 1024, 256, T, 0, n))
        extra = {}
        if dt == torch.float32:
            # float32 ARITHMETIC (stft_pytorch's transform, dvae_stft_f32): bytes = 256 new fp32 samples in + 513 complex64 (or float32) out
            t_cf = timeit(lambda: H.stft_device_f32(x, 1024, 256, T, 2))
            t_pf = timeit(lambda: H.stft_device_f32(x, 1024, 256, T, 1))
            # ... and back (istft_pytorch's transform, dvae_istft_f32): 513 complex64 in + 256 fp32 samples out per frame
            t_if = timeit(lambda: H.istft_device_f32(Sr, 1024, 256, T, 0, n))
            t_if0 = timeit(lambda: H.istft_device_f32(S, 1024, 256, T, 0, n))
            extra = dict(istft_f32arith_us=t_if * 1e6, istft_f32arith_bin_major_us=t_if0 * 1e6, istft_f32arith_GBs=T * (4104 + 1024) / t_if / 1e9,
                         istft_f32arith_hbm_frac=T * (4104 + 1024) / t_if / 8e12,
                         stft_f32arith_us=t_cf * 1e6, stft_f32arith_power_us=t_pf * 1e6, stft_f32arith_GBs=T * (1024 + 4104) / t_cf / 1e9,
                         stft_f32arith_power_GBs=T * (1024 + 2052) / t_pf / 1e9, stft_f32arith_hbm_frac=T * (1024 + 4104) / t_cf / 8e12,
                         stft_f32arith_power_hbm_frac=T * (1024 + 2052) / t_pf / 8e12)
        inb = 256 * x.element_size()
        res[f"{secs}s_{str(dt).split('.')[-1]}"] = dict(frames=T, stft_us=t_c * 1e6, stft_bin_major_us=t_c0 * 1e6, stft_power_us=t_p * 1e6,
            istft_us=t_i * 1e6, istft_bin_major_us=t_i0 * 1e6,
            stft_Mframes_s=T / t_c / 1e6, stft_GBs=T * (inb + 4104) / t_c / 1e9, istft_Mframes_s=T / t_i / 1e6,
            istft_GBs=T * (4104 + 1024) / t_i / 1e9, **extra)
    xs = np.random.default_rng(0).standard_normal(16000 * 60)
    from scipy.signal import get_window
    win = get_window("hann", 1024, fftbins=True)
    t0 = time.perf_counter()
    Tn = 1 + (len(xs) - 1024) // 256
    fr = np.lib.stride_tricks.as_strided(xs, shape=(Tn, 1024), strides=(256 * xs.strides[0], xs.strides[0]))
    np.fft.rfft(fr * win, axis=1).astype(np.complex64)
    t_np = time.perf_counter() - t0
    res["cpu_numpy_rfft_Mframes_s"] = Tn / t_np / 1e6
    print(json.dumps(res, indent=1))

if __name__ == "__main__":
    main()
