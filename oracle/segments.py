"""CPU ORACLE -- TEST INFRASTRUCTURE ONLY.

Numpy stand-ins for the three phases of the segment protocol (pssgp/distributed.py), composed
from the reference restatements in np_oracle.py.  They produce and consume records in exactly
the layout libpgps uses (include/pgps.h, "one series sharded over several GPUs"), so the same
`run_protocol` driver can be exercised on the CPU (gloo, world_size 2) and the GPU records can
be checked field by field.

The reference has no multi-device path; what pins these stand-ins is that stitching the
segments must reproduce the unsegmented `pkf` / `pks` of np_oracle.py (tests/test_segments.py).
"""
import numpy as np

from . import np_oracle as O


def _sym_pack(S):
    d = S.shape[0]
    return np.array([0.5 * (S[i, j] + S[j, i]) for i in range(d) for j in range(i, d)])


def _sym_unpack(v, d):
    S = np.zeros((d, d))
    k = 0
    for i in range(d):
        for j in range(i, d):
            S[i, j] = S[j, i] = v[k]
            k += 1
    return S


def pack_filter_record(elem, F0, Q0):
    A, b, C, J, eta = elem
    return np.concatenate([A.ravel(), b, _sym_pack(C), _sym_pack(J), eta, F0.ravel(), Q0.ravel()])


def unpack_filter_record(rec, d):
    sym = d * (d + 1) // 2
    o = 0
    A = rec[o:o + d * d].reshape(d, d); o += d * d
    b = rec[o:o + d]; o += d
    C = _sym_unpack(rec[o:o + sym], d); o += sym
    J = _sym_unpack(rec[o:o + sym], d); o += sym
    eta = rec[o:o + d]; o += d
    F0 = rec[o:o + d * d].reshape(d, d); o += d * d
    Q0 = rec[o:o + d * d].reshape(d, d)
    return (A, b, C, J, eta), F0, Q0


def pack_smoother_record(elem, ll):
    E, g, L = elem
    body = np.concatenate([E.ravel(), g, _sym_pack(L)])
    if body.size % 2:
        body = np.concatenate([body, [0.0]])
    return np.concatenate([body, [ll, 0.0]])


def unpack_smoother_record(rec, d):
    sym = d * (d + 1) // 2
    n = d * d + d + sym
    pad = n + (n & 1)
    E = rec[:d * d].reshape(d, d)
    g = rec[d * d:d * d + d]
    L = _sym_unpack(rec[d * d + d:n], d)
    return (E, g, L), float(rec[pad])


def _batch(elem):
    return tuple(np.asarray(e)[None] for e in elem)


def _unbatch(elem):
    return tuple(e[0] for e in elem)


class OracleSegment:
    """One rank's segment: (P0, Fs, Qs, H, R) restricted to its steps, ys likewise."""

    def __init__(self, rank, nranks, lgssm_segment, ys):
        self.rank, self.nranks = rank, nranks
        self.P0, self.Fs, self.Qs, self.H, self.R = (np.asarray(a, np.float64) for a in lgssm_segment)
        self.ys = np.asarray(ys, np.float64).reshape(-1)
        self.d = self.Fs.shape[1]

    # phase 1 ---------------------------------------------------------------------------------
    def phase_reduce(self):
        m0 = np.zeros(self.d)
        if self.rank == 0:
            elems = O.make_associative_filtering_elements(m0, self.P0, self.Fs, self.Qs, self.H, self.R, self.ys)
        else:
            elems = O.generic_filtering_elements(self.Fs, self.Qs, self.H, self.R, self.ys)
        self._elems = elems
        total = tuple(e[-1] for e in O.scan_associative(O.filtering_operator, elems))
        return pack_filter_record(total, self.Fs[0], self.Qs[0])

    # phase 2 ---------------------------------------------------------------------------------
    def phase_filter(self, gathered_f):
        d = self.d
        gathered_f = np.asarray(gathered_f, np.float64)
        # carry-in: the prior pushed through the totals of the ranks to the left
        carry = (np.zeros((d, d)), np.zeros(d), self.P0.copy(), np.zeros((d, d)), np.zeros(d))
        for r in range(self.rank):
            tot, _, _ = unpack_filter_record(gathered_f[r], d)
            carry = _unbatch(O.filtering_operator(_batch(carry), _batch(tot)))
        elems = self._elems
        if self.rank > 0:
            first = _unbatch(O.filtering_operator(_batch(carry), tuple(e[0:1] for e in elems)))
            elems = tuple(e.copy() for e in elems)
            for arr, f in zip(elems, first):
                arr[0] = f
        final = O.scan_associative(O.filtering_operator, elems)
        self.fms, self.fPs = final[1], final[2]
        # log-likelihood terms of this segment (np_oracle.pkf, parallel.py:135-151)
        h = self.H.reshape(d)
        r = float(self.R.reshape(()))
        pm = np.concatenate([carry[1][None], self.fms[:-1]], axis=0)
        pP = np.concatenate([carry[2][None], self.fPs[:-1]], axis=0)
        mp = np.einsum("nij,nj->ni", self.Fs, pm)
        Pp = self.Fs @ pP @ np.swapaxes(self.Fs, 1, 2) + self.Qs
        mu = mp @ h
        s2 = np.einsum("i,nij,j->n", h, Pp, h) + r
        lp = -0.5 * (O.LOG2PI + np.log(s2) + (self.ys - mu) ** 2 / s2)
        self.ll_part = float(np.sum(np.where(np.isnan(lp), 0.0, lp)))
        # smoothing elements: the last step of the segment needs F, Q of the next segment's first step
        if self.rank + 1 < self.nranks:
            _, Fn, Qn = unpack_filter_record(gathered_f[self.rank + 1], d)
            Fs = np.concatenate([self.Fs, Fn[None]], axis=0)
            Qs = np.concatenate([self.Qs, Qn[None]], axis=0)
            fm = np.concatenate([self.fms, self.fms[-1:]], axis=0)        # dummy tail row
            fP = np.concatenate([self.fPs, self.fPs[-1:]], axis=0)
            E, g, L = O.make_associative_smoothing_elements(Fs, Qs, fm, fP)
            selems = (E[:-1], g[:-1], L[:-1])
        else:
            selems = O.make_associative_smoothing_elements(self.Fs, self.Qs, self.fms, self.fPs)
        self._selems = selems
        rev = tuple(e[::-1].copy() for e in selems)
        tot = tuple(e[-1] for e in O.scan_associative(O.smoothing_operator, rev))
        return pack_smoother_record(tot, self.ll_part)

    # phase 3 ---------------------------------------------------------------------------------
    def phase_smoother(self, gathered_s):
        d = self.d
        gathered_s = np.asarray(gathered_s, np.float64)
        ll = sum(unpack_smoother_record(gathered_s[r], d)[1] for r in range(self.nranks))
        # (sm, sP) of the first step of the next segment = fold of the totals to the right
        carry = None
        for r in range(self.nranks - 1, self.rank, -1):
            tot, _ = unpack_smoother_record(gathered_s[r], d)
            carry = tot if carry is None else _unbatch(O.smoothing_operator(_batch(carry), _batch(tot)))
        selems = self._selems
        rev = tuple(e[::-1].copy() for e in selems)
        if carry is not None:
            first = _unbatch(O.smoothing_operator(_batch(carry), tuple(e[0:1] for e in rev)))
            for arr, f in zip(rev, first):
                arr[0] = f
        final = O.scan_associative(O.smoothing_operator, rev)
        self.sms, self.sPs = final[1][::-1].copy(), final[2][::-1].copy()
        self.ll = ll
        return ll
