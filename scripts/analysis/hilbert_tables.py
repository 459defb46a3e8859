"""Generates csrc/hilbert.h and tests/hilbert_ref.py: the 3-D Hilbert curve as a finite-state relabelling of
octant digits, derived by exploring Skilling's transpose algorithm on a 64^3 grid, then verified (forward and
inverse) on 200 000 random 20-level points.  python scripts/analysis/hilbert_tables.py"""
import numpy as np, sys
sys.path.insert(0,'/root/repo/scripts/analysis')

def hilbert_index(ix, iy, iz, bits):
    X = [ix.astype(np.uint64).copy(), iy.astype(np.uint64).copy(), iz.astype(np.uint64).copy()]
    M = np.uint64(1) << np.uint64(bits - 1)
    Q = M
    while Q > 1:
        P = Q - np.uint64(1)
        for i in range(3):
            m = (X[i] & Q) != 0
            X[0] = np.where(m, X[0] ^ P, X[0])
            t = (X[0] ^ X[i]) & P
            X[0] = np.where(m, X[0], X[0] ^ t)
            X[i] = np.where(m, X[i], X[i] ^ t)
        Q >>= np.uint64(1)
    for i in range(1, 3):
        X[i] ^= X[i - 1]
    t = np.zeros_like(X[0])
    Q = M
    while Q > 1:
        t = np.where((X[2] & Q) != 0, t ^ (Q - np.uint64(1)), t)
        Q >>= np.uint64(1)
    for i in range(3):
        X[i] ^= t
    h = np.zeros_like(X[0])
    for b in range(bits - 1, -1, -1):
        for i in range(3):
            h = (h << np.uint64(1)) | ((X[i] >> np.uint64(b)) & np.uint64(1))
    return h

bits = 6
g = np.arange(1 << bits)
X, Y, Z = np.meshgrid(g, g, g, indexing='ij')
x, y, z = X.ravel(), Y.ravel(), Z.ravel()
h = hilbert_index(x, y, z, bits).astype(np.int64)
# octant digits (Morton: bx | by<<1 | bz<<2) per level, and Hilbert digits per level
def digit(v, l):  # bit of coordinate at level l (0 = top)
    return (v >> (bits - 1 - l)) & 1
oct_d = [digit(x, l) | (digit(y, l) << 1) | (digit(z, l) << 2) for l in range(bits)]
hil_d = [(h >> (3 * (bits - 1 - l))) & 7 for l in range(bits)]
# state of a cell = the map octant -> hilbert digit among its children (a permutation of 8)
states = {}   # perm tuple -> id
trans = {}    # (state id, octant) -> child state id
perm_of = {}
def cell_perm(mask, l):
    # mask selects the points of one cell at level l (its children are at level l)
    p = [-1] * 8
    for o in range(8):
        sel = mask & (oct_d[l] == o)
        d = np.unique(hil_d[l][sel])
        assert len(d) == 1
        p[o] = int(d[0])
    return tuple(p)
def sid(p):
    if p not in states:
        states[p] = len(states)
    return states[p]
# walk the cell tree breadth first to depth bits-1
front = [(np.ones(len(x), dtype=bool), 0)]
for l in range(bits - 1):
    nxt = []
    for mask, _ in front:
        p = cell_perm(mask, l); s = sid(p)
        for o in range(8):
            cm = mask & (oct_d[l] == o)
            cp = cell_perm(cm, l + 1); cs = sid(cp)
            key = (s, o)
            if key in trans: assert trans[key] == cs, "not a finite-state curve?"
            trans[key] = cs
            nxt.append((cm, 0))
    front = nxt
    if l >= 3: break
print("states", len(states))
S = len(states)
missing = [(s, o) for s in range(S) for o in range(8) if (s, o) not in trans]
print("missing transitions", len(missing))
inv = {v: k for k, v in states.items()}
T = [[inv[s][o] for o in range(8)] for s in range(S)]
N = [[trans.get((s, o), -1) for o in range(8)] for s in range(S)]
print("root state", states[cell_perm(np.ones(len(x), dtype=bool), 0)])

# ---- packed tables, verification, emission ------------------------------------------------------------
import os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

S = len(T)
INV = [[T[s].index(d) for d in range(8)] for s in range(S)]          # digit -> octant
# packed rows: digit row (3 bits per octant), next-state row (5 bits per octant), inverse row (3 bits per digit),
# next-state by DIGIT (5 bits per digit)
def pack(vals, w): 
    r = 0
    for i, v in enumerate(vals): r |= v << (w * i)
    return r
rowD = [pack(T[s], 3) for s in range(S)]
rowN = [pack(N[s], 5) for s in range(S)]
rowI = [pack(INV[s], 3) for s in range(S)]
rowM = [pack([N[s][INV[s][d]] for d in range(8)], 5) for s in range(S)]
# verify against Skilling on random points, 20 levels
rng = np.random.default_rng(0)
bits = 20
P = rng.integers(0, 1 << bits, size=(200000, 3))
h = hilbert_index(P[:, 0], P[:, 1], P[:, 2], bits).astype(np.uint64)
st = np.zeros(len(P), dtype=np.int64); key = np.zeros(len(P), dtype=np.uint64)
rd, rn = np.array(rowD, dtype=np.uint64), np.array(rowN, dtype=np.uint64)
for l in range(bits):
    o = ((P[:, 0] >> (bits - 1 - l)) & 1) | (((P[:, 1] >> (bits - 1 - l)) & 1) << 1) | (((P[:, 2] >> (bits - 1 - l)) & 1) << 2)
    d = (rd[st] >> (3 * o).astype(np.uint64)) & np.uint64(7)
    key = (key << np.uint64(3)) | d
    st = ((rn[st] >> (5 * o).astype(np.uint64)) & np.uint64(31)).astype(np.int64)
assert np.array_equal(key, h), "table recurrence != Skilling"
# and the inverse
st = np.zeros(len(P), dtype=np.int64); mk = np.zeros(len(P), dtype=np.uint64)
ri, rm = np.array(rowI, dtype=np.uint64), np.array(rowM, dtype=np.uint64)
for l in range(bits):
    d = (h >> np.uint64(3 * (bits - 1 - l))) & np.uint64(7)
    o = (ri[st] >> (np.uint64(3) * d)) & np.uint64(7)
    mk = (mk << np.uint64(3)) | o
    st = ((rm[st] >> (np.uint64(5) * d)) & np.uint64(31)).astype(np.int64)
mort = np.zeros(len(P), dtype=np.uint64)
for l in range(bits):
    o = ((P[:, 0] >> (bits - 1 - l)) & 1) | (((P[:, 1] >> (bits - 1 - l)) & 1) << 1) | (((P[:, 2] >> (bits - 1 - l)) & 1) << 2)
    mort = (mort << np.uint64(3)) | o.astype(np.uint64)
assert np.array_equal(mk, mort), "inverse failed"
print("tables verified on", len(P), "points,", bits, "levels")
hdr = '''// hilbert.h - the 3-D Hilbert curve as a finite-state relabelling of octant digits (generated by
// scripts/analysis/hilbert_tables.py from Skilling's transpose algorithm; 24 orientations).
// At a cell in orientation s the child in octant o (x>=cx | (y>=cy)<<1 | (z>=cz)<<2) gets the digit
// (kHilDigit[s] >> 3 o) & 7 and the orientation (kHilNext[s] >> 5 o) & 31; kHilOctant / kHilNextByDigit are the
// same maps indexed by the digit (decoding).  Two bodies share the first L digits iff they share the level-L cell,
// exactly as with the octant digits themselves - only the ORDER of the eight children of a cell changes.
#pragma once
#include <stdint.h>
namespace nbmi {
constexpr int kHilStates = %d;
__device__ __constant__ const uint32_t kHilDigit[kHilStates] = {%s};
__device__ __constant__ const uint64_t kHilNext[kHilStates] = {%s};
__device__ __constant__ const uint32_t kHilOctant[kHilStates] = {%s};
__device__ __constant__ const uint64_t kHilNextByDigit[kHilStates] = {%s};
}  // namespace nbmi
''' % (S, ", ".join("0x%06xu" % v for v in rowD), ", ".join("0x%010xull" % v for v in rowN),
       ", ".join("0x%06xu" % v for v in rowI), ", ".join("0x%010xull" % v for v in rowM))
open(os.path.join(ROOT, '3d-spatial-sim-for-boid-and-nbody_amd', 'csrc', 'hilbert.h'), 'w').write(hdr)
py = '''"""The device's sort key from the reference's octant-path key (test-side mirror of csrc/hilbert.h; generated by
scripts/analysis/hilbert_tables.py).  hilbert_keys(hi, lo): the 2 x 21 octant digits relabelled along the 3-D
Hilbert curve, the state carried from the upper into the lower word."""
import numpy as np

DIGIT = np.array(%r, dtype=np.uint64)
NEXT = np.array(%r, dtype=np.uint64)


def hilbert_keys(hi, lo):
    hi, lo = np.asarray(hi, dtype=np.uint64), np.asarray(lo, dtype=np.uint64)
    st = np.zeros(len(hi), dtype=np.int64)
    out = []
    for word in (hi, lo):
        k = np.zeros(len(hi), dtype=np.uint64)
        for l in range(21):
            o = (word >> np.uint64(3 * (20 - l))) & np.uint64(7)
            k = (k << np.uint64(3)) | ((DIGIT[st] >> (np.uint64(3) * o)) & np.uint64(7))
            st = ((NEXT[st] >> (np.uint64(5) * o)) & np.uint64(31)).astype(np.int64)
        out.append(k)
    return out[0], out[1]
''' % (rowD, rowN)
open(os.path.join(ROOT, 'tests', 'hilbert_ref.py'), 'w').write(py)
