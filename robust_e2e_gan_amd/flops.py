"""Executed matrix-core FLOPs of the entry points a step calls (bench.py ``step_executed_*``).

``lib.call`` adds ``EXECUTED[name](*args)`` to ``lib.FLOP_METER[name]`` while a meter is installed (``with flops.meter() as m:``).  What is counted
is what the MFMA pipes are asked to do by the launch the step REALLY makes: products over the valid rows of a ragged batch count those rows
(re2e_gemm_nt_rows / _tn_rows), row-limited Winograd launches count the rows below their limits, Winograd launches count the transformed-domain
products (F(2x2,3x3): 16 multiply-adds per 2x2 outputs instead of 36; F(2x2,4x4): 25 instead of 64), recurrences count h W_hh^T per live-or-not
(t, b) (the chains multiply all B rows at every step).  Not counted (together < 0.5 % of a config-4 step): the decoder loop's per-token products,
AttLoc, the tile padding of edge tiles.  Cross-check: profiles/r06_step_pmc.json (SQ_VALU_MFMA_BUSY_CYCLES x 64 FLOP per cycle and SIMD)."""
import contextlib

from . import lib

ROW_HINTS = {}        # device address of an int32 row-limit array -> host tuple (ops registers it in front of a row-limited call)


def note_rows(ptr, values):
    if lib.FLOP_METER is not None:
        ROW_HINTS[int(ptr)] = tuple(int(v) for v in values)


def _rows(ptr, n, H, patch=8):
    """sum over the images of the rows a row-limited 3x3 launch computes: whole patches that START below the limit."""
    lim = ROW_HINTS.get(int(ptr)) if ptr is not None else None
    if lim is None:
        return n * H
    return sum(min(H, (min(l, H) + patch - 1) // patch * patch) for l in lim[:n])


def _ptr(p):
    return getattr(p, 'value', p)


def _wino33(a, rows=None):          # (in, NI, H, W, C, w, Cout, ...)
    NI, H, W, C, Cout = a[1], a[2], a[3], a[4], a[6]
    return 2.0 * 4 * C * Cout * (NI * H if rows is None else rows) * W


def _wino44(NI, H, W, C, Cout, pad):
    OH, OW = H + 2 * pad - 3, W + 2 * pad - 3
    return 2.0 * 25 * C * Cout * NI * ((OH + 1) // 2) * ((OW + 1) // 2)


EXECUTED = {
    're2e_gemm': lambda *a: 2.0 * a[2] * a[3] * a[4],
    're2e_gemm_nt_rows': lambda *a: 2.0 * a[0] * a[1] * a[2],
    're2e_gemm_tn_rows': lambda *a: 2.0 * a[0] * a[1] * a[2],
    're2e_gemm_skinny2': lambda *a: 2.0 * a[0] * a[1] * (a[6] + a[11]),
    # (in, NI, H, W, C, wg, Cout, KH, KW, PH, PW, ...)
    're2e_conv_igemm': lambda *a: 2.0 * a[1] * a[9] * a[10] * a[6] * a[7] * a[8] * a[4],
    're2e_conv_igemm_masked': lambda *a: 2.0 * a[1] * a[9] * a[10] * a[6] * a[7] * a[8] * a[4],
    're2e_conv3x3_relu_pool': lambda *a: 2.0 * 9 * a[1] * a[2] * a[3] * a[4] * a[6],
    're2e_conv3x3_wino': lambda *a: _wino33(a),
    're2e_conv3x3_wino_wgrad': lambda *a: _wino33(a),
    're2e_conv3x3_wino_rows': lambda *a: _wino33(a, _rows(_ptr(a[14]), a[1], a[2])),
    're2e_conv3x3_wino_wgrad_rows': lambda *a: _wino33(a, _rows(_ptr(a[9]), a[1], a[2])),
    # (in, NI, H, W, C, w, Cout, pad, dgrad, ...) / (in, NI, H, W, C, dout, Cout, pad, ...)
    're2e_conv4x4_wino': lambda *a: _wino44(a[1], a[2], a[3], a[4], a[6], a[7]),
    're2e_conv4x4_wino_wgrad': lambda *a: _wino44(a[1], a[2], a[3], a[4], a[6], a[7]),
    # (in, NI, H, W, C, dout, Cout, KH, KW, PH, PW, ...)
    're2e_conv_wgrad': lambda *a: 2.0 * a[1] * a[9] * a[10] * a[6] * a[7] * a[8] * a[4],
    # (dz, N, OH, OW, Cout, W, Cin, KH, KW, H, Wd, pad, ...): every input pixel gathers KH*KW/4 taps
    're2e_conv_dgrad_s2': lambda *a: 2.0 * a[1] * a[9] * a[10] * a[6] * a[4] * (a[7] * a[8] / 4.0),
    # (xg_f, xg_r, whh_f, whh_r, ybuf, cbuf, lens, T, B, H, ...): h W_hh^T, both directions
    're2e_lstm_seq_fwd': lambda *a: 2.0 * 2 * a[7] * a[8] * a[9] * 4 * a[9],
    # (g_f, g_r, whh_f, whh_r, dy, ybuf, cbuf, dc, lens, T, B, H, ...): d(gates) W_hh
    're2e_lstm_seq_bwd': lambda *a: 2.0 * 2 * a[9] * a[10] * a[11] * 4 * a[11],
    # (x, rows, F, NF, ...): (x^2) W on the matrix cores (dense form: the band structure is not exploited there)
    're2e_fbank_fwd': lambda *a: 2.0 * a[1] * a[2] * a[3],
    're2e_fbank_bwd': lambda *a: 2.0 * a[1] * a[2] * a[3],
}


@contextlib.contextmanager
def meter():
    """``with meter() as m:`` -- m[name] = executed FLOPs of the successful calls of ``name`` inside the block (host side, no device work)."""
    prev, lib.FLOP_METER = lib.FLOP_METER, {}
    try:
        yield lib.FLOP_METER
    finally:
        lib.FLOP_METER = prev
        ROW_HINTS.clear()


# direct-form FLOPs of the same launch / executed FLOPs: what a Winograd launch replaces (SURVEY 8(d) counts the direct form)
DIRECT_OVER_EXECUTED = {'re2e_conv3x3_wino': 2.25, 're2e_conv3x3_wino_rows': 2.25, 're2e_conv3x3_wino_wgrad': 2.25, 're2e_conv3x3_wino_wgrad_rows': 2.25,
                        're2e_conv4x4_wino': 64.0 / 25.0, 're2e_conv4x4_wino_wgrad': 64.0 / 25.0}
DIRECT = '(direct-form equivalent of all of the above)'


def count(name, args):
    fn = EXECUTED.get(name)
    if fn is not None:
        m = lib.FLOP_METER
        f = float(fn(*args))
        m[name] = m.get(name, 0.0) + f
        m[DIRECT] = m.get(DIRECT, 0.0) + f * DIRECT_OVER_EXECUTED.get(name, 1.0)


def totals(m):
    """(executed FLOPs, direct-form equivalent FLOPs) of a meter"""
    return sum(v for k, v in m.items() if k != DIRECT), m.get(DIRECT, 0.0)
