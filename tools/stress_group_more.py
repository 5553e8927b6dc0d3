"""tools/stress_group_more.py [first_seed] [count] — the random-graph parity checks of tests/test_stress_group_gpu.py on many more
seeds than the suite carries (development aid; every group kernel family, widths 12..192, 2..16 slices, value-free and weighted,
bias + ReLU epilogue, bitwise repeatability), against the fp64 oracle.  Exits non-zero at the first mismatch."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import gcn_amd                                          # noqa: E402
from test_stress_group_gpu import _graph, TOL           # noqa: E402
from util import oracle_spmm, rel_err                   # noqa: E402


def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 150
    d = torch.device("cuda:0")
    seen = {}
    worst = 0.0
    for seed in range(first, first + count):
        n, rowptr, col, val, rng = _graph(seed)
        S = int(rng.choice([2, 3, 4, 5, 8, 11, 13, 15, 16]))
        if seed % 4 == 3:
            val = (val * (1.0 + 0.5 * rng.random(len(val)))).astype(np.float32)
        adj = gcn_amd.CsrAdjacency(torch.from_numpy(rowptr).to(d), torch.from_numpy(col).to(d), torch.from_numpy(val).to(d),
                                   (n, n), slices=S)
        for k in (int(rng.choice([33, 36, 40, 41, 44, 45, 47, 48])), int(rng.choice([12, 16, 20, 24, 28, 32])),
                  int(rng.choice([52, 64, 100, 128, 172, 192]))):
            name = adj.main_kernel(k).split("<")[0]
            seen[name] = seen.get(name, 0) + 1
            B = rng.standard_normal((n, k)).astype(np.float32)
            ref = oracle_spmm(rowptr, col, val, B)
            Bd = torch.from_numpy(B).to(d)
            C = adj.matmul_raw(Bd)
            err = rel_err(C.cpu().numpy(), ref)
            worst = max(worst, err)
            if not (err <= TOL) or not torch.equal(C, adj.matmul_raw(Bd)):
                print("MISMATCH", seed, n, S, k, name, err)
                sys.exit(1)
            bias = rng.standard_normal(k).astype(np.float32)
            Ce = adj.matmul_raw(Bd, bias=torch.from_numpy(bias).to(d), relu=True).cpu().numpy()
            if not (rel_err(Ce, np.maximum(ref + bias, 0)) <= TOL):
                print("MISMATCH (epilogue)", seed, n, S, k, name)
                sys.exit(1)
    print("ok: seeds %d..%d, worst rel err %.2e, kernels %s" % (first, first + count - 1, worst, dict(sorted(seen.items()))))


if __name__ == "__main__":
    main()
