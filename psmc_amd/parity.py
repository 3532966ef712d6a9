"""What "fast mode agrees with exact mode" means, in numbers (tests/ and bench.py; not part of the E-step path).

The gate of the fast kernels has been max|x - ref| / max|ref| <= 1e-10 over the whole count matrix -- silent about small cells.
The M-step reads the counts through hmm_Q (khmm.c:363-382): Q = sum_kl A[k][l] log a[k][l] + sum_bk E[b][k] log e[b][k], so what
matters downstream is (i) the element-wise relative error of the cells that carry weight and (ii) the error of those two sums.
"""
import numpy as np


def fast_error_metrics(r, o, a=None, e=None, floor=1e-6):
    """r, o: dicts with A (n, n), E (2, n), LL -- a fast result and its exact / oracle reference.  Returns
    A_max, E_max   max |x - ref| / max |ref|                      (the gate since round 1)
    A_cell, E_cell largest RELATIVE error over the cells >= floor x the largest cell
    A_l1           sum |A - ref| / sum |ref|: bounds the error of any linear functional sum w A by max |w| x this x sum |ref|
    LL             relative
    QA, QE         (when a, e are given) relative error of sum A log a and of sum E log e, the two sums hmm_Q consumes"""
    A, Ao = np.asarray(r["A"], float), np.asarray(o["A"], float)
    E, Eo = np.asarray(r["E"], float)[:2], np.asarray(o["E"], float)[:2]
    m = dict(A_max=float(np.abs(A - Ao).max() / np.abs(Ao).max()), E_max=float(np.abs(E - Eo).max() / np.abs(Eo).max()))
    big = Ao >= floor * Ao.max()
    m["A_cell"] = float((np.abs(A - Ao)[big] / Ao[big]).max())
    bigE = Eo >= floor * Eo.max()
    m["E_cell"] = float((np.abs(E - Eo)[bigE] / Eo[bigE]).max())
    m["A_l1"] = float(np.abs(A - Ao).sum() / np.abs(Ao).sum())
    m["LL"] = float(abs(r["LL"] - o["LL"]) / abs(o["LL"]))
    if a is not None and e is not None:
        a = np.asarray(a, float); e2 = np.asarray(e, float)[:2]
        pa, pe = a > 0, e2 > 0
        qa_ref, qe_ref = float((Ao[pa] * np.log(a[pa])).sum()), float((Eo[pe] * np.log(e2[pe])).sum())
        m["QA"] = abs(float((A[pa] * np.log(a[pa])).sum()) - qa_ref) / abs(qa_ref)
        m["QE"] = abs(float((E[pe] * np.log(e2[pe])).sum()) - qe_ref) / abs(qe_ref)
    return m
