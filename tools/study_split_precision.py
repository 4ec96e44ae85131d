#!/usr/bin/env python3
"""Accuracy of operand splits for an fp32-accurate contraction on 16-bit matrix cores (CPU study, numpy + torch).

    python tools/study_split_precision.py

Compares, on dot products shaped like the path's (K = 432, He-scaled weights), against fp64:
  fp32     a plain fp32 GEMM,
  bf16x3   what igemm4 / fcgemm execute today: hi+mid+lo bf16 pieces, the six products >= 2^-16, fp32 accumulate,
  fp16x2   DESIGN.md section 11, candidate 1: x*s = h + l * 2^-11 with fp16 h, l and a power-of-two scale s per
           tensor (max |x*s| <= 2^14), products h.h into one accumulator and h.l + l.h into a second one that is
           added with the factor 2^-11: three MFMAs instead of six, 4 instead of 6 bytes per staged element.
Errors are relative to sum |terms| of each dot product (what an fp32 summation error is measured against).
"""
import numpy as np
import torch


def bf16(x):
    return torch.from_numpy(x).to(torch.bfloat16).to(torch.float32).numpy()


def f16(x):
    return x.astype(np.float16).astype(np.float32)


def split16(x):
    s = 2.0 ** (14 - np.ceil(np.log2(np.abs(x).max())))
    xs = (x * s).astype(np.float32)
    h = f16(xs)
    return h, f16((xs - h) * 2048.0), s


def main():
    rs = np.random.RandomState(0)
    M, K, N = 4096, 432, 8
    cases = (('ReLU of N(0,1)', np.maximum(rs.randn(M, K), 0)),
             ('1e-6 * lognormal(3)', rs.randn(M, K) * 1e-6 * np.exp(3 * rs.randn(M, K))),
             ('lognormal(4)', rs.randn(M, K) * np.exp(4 * rs.randn(M, K))))
    for name, a in cases:
        a = a.astype(np.float32)
        w = (rs.randn(K, N) * np.sqrt(2 / K)).astype(np.float32)
        ref = a.astype(np.float64) @ w.astype(np.float64)
        scale = np.abs(a.astype(np.float64)) @ np.abs(w.astype(np.float64))
        ah = bf16(a); am = bf16(a - ah); al = bf16(a - ah - am)
        wh = bf16(w); wm = bf16(w - wh); wl = bf16(w - wh - wm)
        b3 = (ah @ wh + (ah @ wm + am @ wh) + (ah @ wl + al @ wh + am @ wm)).astype(np.float32)
        a_h, a_l, sa = split16(a)
        w_h, w_l, sw = split16(w)
        h2 = (((a_h @ w_h).astype(np.float32) + (a_h @ w_l + a_l @ w_h).astype(np.float32) / 2048.0) / (sa * sw)).astype(np.float32)
        for nm, v in (('fp32', a @ w), ('bf16x3', b3), ('fp16x2', h2)):
            e = np.abs(v - ref) / scale
            print('%-20s %-7s max %.2e  rms %.2e' % (name, nm, e.max(), np.sqrt((e ** 2).mean())))


if __name__ == '__main__':
    main()
