#!/usr/bin/env python3
"""torch's CPU float32 sum over a contiguous inner dimension, restated element by element (aten/src/ATen/native/cpu/
SumKernel.cpp of torch 2.10: vectorized_inner_sum -> row_sum -> multi_row_sum; 8-float vectors, 4 interleaved streams, a
4-level cascade with 16-element level-0 chains, scalar tail first, then the 8 lanes in order) and checked against torch.sum
bit for bit for row lengths 8 .. 20 000 (run it: CPU only).  This is the order of the reference's row sums in
xie_propagation_points_in_order (/root/reference field_utils.py:590-595: torch.sum(interaction_mat[idx] * weights, dim=-1)).
Round 4 asked whether the ordered-propagation kernel should sum in this order instead of fp64.  Answer: it could (32
independent chains per row, level-0 blocks in parallel), but it would not make the sums the reference's bits - the
reference's interaction matrix differs from the kernel's (and from any IEEE op-by-op evaluation on the GPU) in 10 % of
its entries by one ulp (tools/gpu_sign_slack.py) - and with fp64 sums the number of decisions that differ from the
reference's goldens is ZERO (profiles/r04_sign_slack.txt).  Kept as the record of what "the reference's order" is."""
import numpy as np, torch, math
f32 = np.float32
def ceil_log2(x):
    return 0 if x <= 1 else int(math.ceil(math.log2(x)))
def multi_row_sum(vecs, nrows):
    # vecs: [size*nrows... ] array of shape [n_vec, V]; stream k takes vectors i*nrows + k, i < size
    size = vecs.shape[0] // nrows
    V = vecs.shape[1]
    num_levels = 4
    level_power = max(4, ceil_log2(size) // num_levels)
    level_step = 1 << level_power
    level_mask = level_step - 1
    acc = np.zeros((num_levels, nrows, V), dtype=f32)
    i = 0
    while i + level_step <= size:
        for j in range(level_step):
            for k in range(nrows):
                acc[0][k] = acc[0][k] + vecs[i * nrows + k]
            i += 1
        for j in range(1, num_levels):
            acc[j] = acc[j] + acc[j - 1]
            acc[j - 1] = 0
            mask = level_mask << (j * level_power)
            if (i & mask) != 0:
                break
    while i < size:
        for k in range(nrows):
            acc[0][k] = acc[0][k] + vecs[i * nrows + k]
        i += 1
    for j in range(1, num_levels):
        acc[0] = acc[0] + acc[j]
    return acc[0]
def torch_cpu_row_sum(row, V=8):
    row = row.astype(f32)
    n = row.shape[0]
    vec_size = n // V
    vecs = row[:vec_size * V].reshape(vec_size, V)
    ilp = 4
    size_ilp = vec_size // ilp
    ps = multi_row_sum(vecs[:size_ilp * ilp], ilp) if size_ilp > 0 else np.zeros((ilp, V), dtype=f32)
    for i in range(size_ilp * ilp, vec_size):
        ps[0] = ps[0] + vecs[i]
    for k in range(1, ilp):
        ps[0] = ps[0] + ps[k]
    final = f32(0)
    for k in range(vec_size * V, n):
        final = f32(final + row[k])
    for l in range(V):
        final = f32(final + ps[0][l])
    return final
if __name__ == "__main__":
    g = torch.Generator().manual_seed(1)
    bad = 0
    for n in (8, 31, 32, 33, 100, 511, 512, 999, 1000, 1001, 2048, 4096, 5000, 10000, 16384, 20000):
        x = (torch.randn(5, n, generator=g) * torch.randn(5, n, generator=g)).contiguous()
        want = torch.sum(x, dim=-1).numpy()
        got = np.array([torch_cpu_row_sum(x[r].numpy()) for r in range(5)], dtype=f32)
        ok = np.array_equal(want.view(np.uint32), got.view(np.uint32))
        bad += not ok
        print(n, "match" if ok else f"DIFFER {want} {got}")
    print("mismatches:", bad)
