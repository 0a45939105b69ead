"""Device-mesh choice and x-slab halo exchange for the multi-GPU path.

factors(N) keeps the role of GB-25 src/sharding_utils.jl:39-62 (the 2:1 device mesh used for weak
scaling).  The single-node configurations of BASELINE.json use an x-slab decomposition instead
(SURVEY.md section 8e): every rank owns Nx/P columns and talks to its west and east neighbour only.
"""
import math


def factors(N):
    """(Dx, Dy) with Dx*Dy == N and Dx == 2*Dy, plus the reference's special cases."""
    special = {4: (2, 2), 16: (4, 4), 512: (32, 32), 6136: (104, 59), 9152: (143, 64), 9180: (135, 68),
               16384: (128, 128)}
    if N in special:
        return special[N]
    if N % 2:
        raise ValueError(f"N must be even; got N = {N}")
    half = N // 2
    D = math.isqrt(half)
    if D * D != half:
        raise ValueError(f"N / 2 = {half} is not a perfect square")
    return 2 * D, D


def slab_neighbours(rank, nranks):
    """(west, east) ranks of an x-slab on the periodic ring."""
    return (rank - 1) % nranks, (rank + 1) % nranks


def mesh_neighbours(rank, Rx, Ry):
    """Neighbours of rank = ry Rx + rx in a Partition(Rx, Ry, 1) mesh: the periodic ring in x within the row, the southern and
    northern rank of the column (None beyond the walls / the fold), the fold partner within the row (mirrored in x)."""
    rx, ry = rank % Rx, rank // Rx
    return dict(west=ry * Rx + (rx - 1) % Rx, east=ry * Rx + (rx + 1) % Rx,
                south=(ry - 1) * Rx + rx if ry > 0 else None, north=(ry + 1) * Rx + rx if ry < Ry - 1 else None,
                partner=ry * Rx + (Rx - 1 - rx))
