"""Random solver configurations: the specialised / tiled kernels against the plain per-pixel kernels of the library
(kernel path 1, itself held to the oracle by the other tests), whole solves compared bit for bit.  Sizes, scales,
window lengths, blur and BTV sizes, channel counts and motion amplitudes are drawn at random, so image edges, ragged
tiles, windows of one frame and motions beyond the forward kernel's LDS window all occur."""
import numpy as np
import pytest


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [101, 202, 303])
def test_specialised_kernels_equal_plain_kernels_on_random_solves(seed):
    import torch
    from superresolution_amd import api, ops

    rng = np.random.default_rng(seed)
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    done = 0
    try:
        for it in range(40):
            scale = int(rng.choice([2, 3, 4, 4, 4]))
            big = rng.random() < 0.2
            rows = int(rng.integers(60, 160)) if big else int(rng.integers(3, 60))
            cols = int(rng.integers(100, 300)) if big else int(rng.integers(3, 90))
            ch = int(rng.choice([1, 3]))
            K = int(rng.integers(1, 12))
            base = int(rng.integers(0, K))
            T = int(rng.integers(1, 4))
            blur = int(rng.choice([5, 5, 5, 3, 7, 1]))
            sigma = float(rng.choice([0.0, 0.0, 1.1]))
            btv = int(rng.choice([7, 7, 5, 3, 9]))
            if min(rows, cols) * scale <= 2 * btv:
                continue
            lam = float(rng.choice([0.03, 0.0, 0.1]))
            amp = float(rng.choice([0.1, 0.4, 1.5, 4.0]))
            yy, xx = np.mgrid[0:rows, 0:cols].astype(np.float32)
            fl = []
            for k in range(K - 1):
                f = np.stack([amp * np.sin(yy / 9 + xx / 13 + it + k), amp * np.cos(yy / 11 - xx / 7 + it - k)], -1)
                f = f + rng.uniform(-amp, amp, 2)
                if rng.random() < 0.3:
                    f = f + rng.standard_normal(f.shape) * 0.5
                fl.append(d(f.astype(np.float32)))
            y = [d((rng.random((rows, cols, ch)) * 255).round().astype(np.float32)) for _ in range(K)]
            outs = []
            for path in (1, 0):
                ops.set_kernel_path(path)
                sr = api.BTV_L1()
                for name, v in (("scale", scale), ("iterations", T), ("blurKernelSize", blur), ("blurSigma", sigma),
                                ("btvKernelSize", btv), ("lambda", lam)):
                    sr.set(name, v)
                out, full = ops.process_device(sr, y, fl, base, return_full=True)
                outs.append((out.cpu().numpy(), full.cpu().numpy()))
            case = dict(scale=scale, rows=rows, cols=cols, ch=ch, K=K, base=base, T=T, blur=blur, sigma=sigma, btv=btv,
                        lam=lam, amp=amp)
            assert np.array_equal(outs[0][1], outs[1][1]) and np.array_equal(outs[0][0], outs[1][0]), case
            done += 1
    finally:
        ops.set_kernel_path(0)
    assert done >= 30
