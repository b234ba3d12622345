#!/usr/bin/env python3
"""LDS bank conflicts of the NTT passes (csrc/ntt.hip), counted by replaying the index arithmetic of every step for the lanes of one
wave: for each of a step's loads / stores, the worst number of lanes of a group that fall on one bank at different addresses.  Two
machine models: 32 lanes x 32 banks per cycle and 64 x 64 (the guide does not say which gfx950 implements; both are printed).

    python tools/lds_conflicts.py            # the layouts of the library, plain and with the index swizzles tried in round 5

Round 5's finding (profiles/r05_experiments.md): swizzles that make every step of the two hot shapes conflict-free in the 32 x 32 model
(below) change NOTHING in the measured time of a 2^20 transform (100.4 vs 99.8 us), while removing the LDS traffic and the barriers
altogether (wrong results, timing only) takes 20 us off: the LDS phases cost their instruction count and the synchronisation, not
bank conflicts.  The library keeps the plain layouts.
"""
import sys


def bitrev(x, bits):
    return int(format(x, f"0{bits}b")[::-1], 2) if bits else 0


def steps_of(log_np):
    s, out = 0, []
    while s < log_np:
        r = 1 if (log_np - s) & 1 else 2
        out.append((s, r))
        s += r
    return out


def pass_indices(log_np, log_t, threads, wave, swz):
    """yields (label, [word index per lane]) for every LDS access of wave `wave` of k_ntt_pass_f"""
    T = 1 << log_t
    for (s, r) in steps_of(log_np):
        first, last = s == 0, s + r == log_np
        hgt = 1 << s
        items = (1 << (log_np + log_t)) >> r
        for it0 in range(wave * 64, items, threads):
            lanes = []
            for l in range(64):
                it = it0 + l
                c, b = it & (T - 1), it >> log_t
                j = b & (hgt - 1)
                lo_d = ((b >> s) << (s + r)) | j
                lanes.append((c, lo_d))
            for k in range(1 << r):
                idx = [swz(((lo_d + k * hgt) << log_t) | c) for (c, lo_d) in lanes]
                if not first:
                    yield (f"S={s} load k={k}", idx)
                if not last:
                    yield (f"S={s} store k={k}", idx)
            break          # one round of the item loop is representative


def last_indices(log_np, log_t, threads, wave, row, swz):
    T = 1 << log_t
    for (s, r) in steps_of(log_np):
        first, last = s == 0, s + r == log_np
        hgt = 1 << s
        items = (1 << (log_np + log_t)) >> r
        for it0 in range(wave * 64, items, threads):
            lanes = []
            for l in range(64):
                it = it0 + l
                if last:
                    c, b = it & (T - 1), it >> log_t
                else:
                    b, c = it & ((1 << (log_np - r)) - 1), it >> (log_np - r)
                j = b & (hgt - 1)
                lo_d = ((b >> s) << (s + r)) | j
                lanes.append((c, lo_d))
            for k in range(1 << r):
                idx = [swz(c, lo_d + k * hgt, row) for (c, lo_d) in lanes]
                if not first:
                    yield (f"S={s} load k={k}", idx)
                if not last:
                    yield (f"S={s} store k={k}", idx)
            break


def cost(idx, group, banks):
    """cycles of one wave-wide 4-byte access: per group of lanes, the largest number of distinct addresses on one bank"""
    total = 0
    for g0 in range(0, 64, group):
        per_bank = {}
        for a in idx[g0:g0 + group]:
            per_bank.setdefault(a % banks, set()).add(a)
        total += max(len(v) for v in per_bank.values())
    return total


def report(name, gen):
    acc = list(gen)
    for group, banks in ((32, 32), (64, 64)):
        cyc = sum(cost(idx, group, banks) for _, idx in acc)
        ideal = len(acc) * (64 // group)
        worst = {}
        for label, idx in acc:
            key = label.rsplit(" k=", 1)[0]
            worst[key] = max(worst.get(key, 0), cost(idx, group, banks) / (64 // group))
        print(f"{name:46s} {group}x{banks}: {cyc:5d} cycles per limb plane (conflict-free {ideal}), worst ways: " + ", ".join(f"{k} {v:.0f}" for k, v in worst.items()))


# ---- the swizzles that were tried (not in the library) ----
def swz_pass(log_t):
    def f(idx):
        d = idx >> log_t
        d ^= ((d >> 3) & 1) | (((d >> 4) & 1) * 6)
        return (d << log_t) | (idx & ((1 << log_t) - 1))
    return f


def swz_last(c, dl, row):
    dl ^= (((dl >> 5) & 1) * 0x05) ^ (((dl >> 6) & 1) * 0x1A) ^ ((c & 1) * 0x10)
    return c * row + dl


if __name__ == "__main__":
    for log_np, log_t in ((10, 2), (9, 3), (11, 1), (8, 4)):
        threads = max(64, min(1024, (1 << (log_np + log_t)) >> 2))
        report(f"pass 2^{log_np} x T={1 << log_t}, plain", pass_indices(log_np, log_t, threads, 1, lambda i: i))
        report(f"pass 2^{log_np} x T={1 << log_t}, swizzled", pass_indices(log_np, log_t, threads, 1, swz_pass(log_t)))
    for log_np, log_t in ((10, 1), (9, 2), (11, 0), (8, 3)):
        threads = max(64, min(1024, (1 << (log_np + log_t)) >> 2))
        pad = 8 if log_np >= 8 else 0
        report(f"last 2^{log_np} x T={1 << log_t}, padded rows (plain)", last_indices(log_np, log_t, threads, 1, (1 << log_np) + pad, lambda c, dl, row: c * row + dl))
        report(f"last 2^{log_np} x T={1 << log_t}, swizzled", last_indices(log_np, log_t, threads, 1, 1 << log_np, swz_last))
