"""Test infrastructure: plain-Python restatements of the reference's HOST helpers on the dipole path, used only
by tests/ to check the product's native / device implementations.  Not imported by the product.

    merge_nodes   util.merge_nodes, util.py:448-492 - the literal double loop over cells (O(cells^2)), including
                  its order dependence (the last touching cell wins, up to 10 sweeps)
"""
from typing import List, Tuple


def merge_nodes(sizes: List[int], ijk: List[Tuple[int, int, int]], min_patch: int):
    """Returns (groups, sweeps): groups[g] = original cell ids of surviving patch g in concatenation order."""
    size = [int(n) for n in sizes]
    members = [[c] for c in range(len(size))]
    cells = [[tuple(c)] for c in ijk]
    live = [True] * len(size)

    def touches(a, b):
        for ca in a:
            for cb in b:
                if abs(ca[0] - cb[0]) <= 1 and abs(ca[1] - cb[1]) <= 1 and abs(ca[2] - cb[2]) <= 1:
                    return True
        return False

    sweeps, again = 0, True
    while again and sweeps < 10:
        again = False
        sweeps += 1
        for i in range(len(cells)):
            if not live[i] or size[i] >= min_patch:
                continue
            target = -1
            for j in range(len(cells)):                     # find_dij keeps overwriting: the LAST match wins
                if j != i and live[j] and touches(cells[i], cells[j]):
                    target = j
            if target < 0:
                continue
            members[target].extend(members[i])
            size[target] += size[i]
            cells[target].extend(cells[i])
            live[i] = False
            members[i], cells[i], size[i] = [], [], 0
            if size[target] < min_patch:
                again = True
    groups = [members[i] for i in range(len(cells)) if live[i] and size[i] >= min_patch]
    return groups, sweeps
