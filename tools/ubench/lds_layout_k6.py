#!/usr/bin/env python3
"""LDS layout search for k_line_products_k (ripp_amd/csrc/fq_line_products_k.hpp).

A ds_read_b128 is served 16 lanes (256 B = the 64 banks once) per cycle; two lanes of one 16-lane phase that want DIFFERENT 16-byte units in the same
bank quad serialise.  Lane l of the wave is (group g = l mod 10, output k = l div 10) -- k-major -- and reads, for term d in (0, 4, 3), slot (k + d) mod 6 of
its group's accumulator: unit g * G + ((k + d) mod 6) * P (+ the same constant for every lane).  The line values are read at unit g * GY (+ D on the lanes
whose term wraps around w^6: k < 2 for term 1, k < 3 for term 2).  Cost = serialised cycles summed over the 3 terms x 4 phases (12 = conflict-free).
Prints the best (G, P) and (GY, D); the kernel uses G = 65, P = 10 (cost 15) and GY = 61, D = 12 (cost 14), 1 260 units = 20 160 B per wave.
For comparison the group-major numbering l = 6 g + k cannot do better than 21."""
def lanes(ph, kmajor=True):
    for l in range(16 * ph, 16 * ph + 16):
        if l >= 60: yield (0, 0)
        elif kmajor: yield (l % 10, l // 10)
        else: yield (l // 6, l % 6)
def cyc(units):
    cnt = {}
    for u in set(units): cnt[u % 16] = cnt.get(u % 16, 0) + 1
    return max(cnt.values())
def cost_x(G, P, kmajor=True):
    return sum(cyc([g * G + ((k + d) % 6) * P for g, k in lanes(ph, kmajor)]) for d in (0, 4, 3) for ph in range(4))
def cost_y(GY, D, kmajor=True):
    return sum(cyc([g * GY + (D if k < thr else 0) for g, k in lanes(ph, kmajor)]) for thr in (0, 2, 3) for ph in range(4))
if __name__ == "__main__":
    for km in (True, False):
        bx = sorted((cost_x(G, P, km), G, P) for P in range(8, 16) for G in range(6 * P, 6 * P + 24))
        by = sorted((cost_y(GY, D, km), GY, D) for GY in range(60, 70) for D in (12,))
        print("k-major" if km else "group-major", "accumulators (cost, G, P):", bx[:4], " line values (cost, GY, D):", by[:3])
