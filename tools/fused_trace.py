"""Summarise PCY_FUSED_TRACE output (wall_clock64 stamps, 100 MHz) of the persistent decode kernel: for every streaming
workgroup and phase [start-of-wait, x staged, all 7 streaming waves done, published]; for every attention workgroup and
layer [start, end].  Usage: fused_trace.py trace.txt [n_stream=192] [n_attn=64]"""
import sys
import numpy as np
t = np.loadtxt(sys.argv[1], dtype=np.float64) / 100.0   # us
NS = int(sys.argv[2]) if len(sys.argv) > 2 else 192
NA = int(sys.argv[3]) if len(sys.argv) > 3 else 64
L = (len(t) // 256 - 4) // 18
P = 4 * L + 1
s = t[:NS * P * 4].reshape(NS, P, 4)
att = t[NS * P * 4:NS * P * 4 + NA * L * 2].reshape(NA, L, 2)
names = ["qkv", "o", "gate/up", "down"]
print(f"step total {s[:, -1, 3].max() - s[:, 0, 0].min():.1f} us over {L} layers, {NS} streaming + {NA} attention workgroups")
print("per phase kind (mean over layers 1..): all times in us")
print("          xready spread | first xready -> last done (phase makespan) | done spread | last publish -> first xready(next)")
for k in range(4):
    idx = np.arange(k, 4 * L, 4)[1:]
    xr, dn, pb = s[:, idx, 1], s[:, idx, 2], s[:, idx, 3]
    nxt = s[:, idx + 1, 1]
    print(f"{names[k]:8s} {np.mean(xr.max(0) - xr.min(0)):6.2f}        | {np.mean(dn.max(0) - xr.min(0)):6.2f}"
          f"                                   | {np.mean(dn.max(0) - dn.min(0)):6.2f}      | {np.mean(nxt.min(0) - pb.max(0)):6.2f}  "
          f"(last publish -> last xready(next) {np.mean(nxt.max(0) - pb.max(0)):6.2f})")
q = np.arange(0, 4 * L, 4)[1:]
print(f"attention: qkv last publish -> first attn start n/a; attn body (start->end) mean {np.mean(att[:, 1:, 1] - att[:, 1:, 0]):.2f}; "
      f"last qkv publish -> last attention end {np.mean(att[:, 1:, 1].max(0) - s[:, q, 3].max(0)):.2f}; "
      f"last attention end -> first o xready {np.mean(s[:, q + 1, 1].min(0) - att[:, 1:, 1].max(0)):.2f}")
