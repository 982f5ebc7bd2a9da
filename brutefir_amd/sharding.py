"""How the crossbar is split over the GPUs of one node (partitioning B).

The reference splits filters over fork()ed processes (bfconf.c:2227-2318) and lets every
process read all input spectra from shared memory; an output is always mixed inside one
process (bfconf.c:2893-2931).  With one process per GPU we shard by INPUT channel instead:
rank r owns a contiguous slice of the inputs, their spectrum rings and the coefficient sets of
every filter fed by them, computes partial output spectra for ALL outputs, and one
reduce-scatter (RCCL over xGMI, sum, channel-major) leaves every rank with the finished spectra
of its slice of the outputs, which it inverse-transforms.  That is the crossbar mix-down
`north_star` asks for: 1 collective per block, O*L complex numbers per rank in, O*L/G out.

Pure index arithmetic -- no device code, usable (and tested) on CPU with gloo.
"""


def split_even(n, parts, index):
    """contiguous [first, first+count) of n items for `index` of `parts` (n % parts == 0 is
    required by reduce_scatter_tensor's equal chunks)"""
    if n % parts != 0:
        raise ValueError("%d channels do not split evenly over %d ranks" % (n, parts))
    count = n // parts
    return index * count, count


def shard_crossbar(n_in, n_out, world_size, rank):
    """returns (first_in, count_in, first_out, count_out) owned by `rank`"""
    fi, ci = split_even(n_in, world_size, rank)
    fo, co = split_even(n_out, world_size, rank)
    return fi, ci, fo, co


def mixdown(z_partial, z_local, group=None):
    """Sum the ranks' partial output spectra and leave each rank its own output slice.

    z_partial: [n_out, L, 2] real view of the complex partial spectra (all outputs)
    z_local:   [n_out / world, L, 2] receives this rank's finished spectra
    Works on any backend torch.distributed offers ("nccl" = RCCL on ROCm; "gloo" on CPU).
    """
    import torch.distributed as dist
    if dist.get_backend(group) == "gloo":
        # gloo has no reduce_scatter: all-reduce on the host, then slice (rehearsal only:
        # CPU tests, or several ranks sharing one GPU where RCCL refuses duplicate devices)
        tmp = z_partial.detach().cpu().clone()
        dist.all_reduce(tmp, group=group)
        n = z_local.shape[0]
        r = dist.get_rank(group)
        z_local.copy_(tmp[r * n:(r + 1) * n])
    else:
        dist.reduce_scatter_tensor(z_local, z_partial, group=group)
    return z_local


def load_balance_filters(filters, n_ranks):
    """The reference's own partition of a filter network (bfconf.c:2227-2318): filters that are
    connected through from_filters / to_filters, or that mix into the same output, must share a
    process; the groups that remain are dealt round-robin over the processes.  Groups built this
    way never exchange OUTPUT data -- every output is finished where it is computed.  For
    configurations like massive_config (independent channels) each GPU simply runs its groups
    with an engine of its own; a crossbar falls apart by output channel, and every process then
    needs the spectra of ALL inputs (shared memory in the reference; an all-gather of input
    spectra between GPUs).  `shard_crossbar` + `mixdown` split by input instead: each GPU
    transforms only its own inputs and one reduce-scatter finishes the outputs.

    filters: list of dicts with optional keys in_f (indices of source filters) and out_ch.
    Returns (rank of every filter, number of ranks actually used)."""
    n = len(filters)
    proc = [-1] * n
    links = [set(f.get("in_f", ())) for f in filters]
    for i, f in enumerate(filters):                       # to_filters is the same edge seen from the source
        for g in f.get("in_f", ()):
            links[g].add(i)
    process = 0
    for first in range(n):
        if proc[first] != -1:
            continue
        proc[first] = process
        changed = True
        while changed:
            changed = False
            for i in range(n):
                if proc[i] != process:
                    continue
                for k in links[i]:
                    if proc[k] != process:
                        proc[k] = process
                        changed = True
            used = set()
            for i in range(n):
                if proc[i] == process:
                    used.update(filters[i].get("out_ch", ()))
            for i in range(n):
                if proc[i] != process and used.intersection(filters[i].get("out_ch", ())):
                    proc[i] = process
                    changed = True
        process += 1
    ranks = [p % n_ranks for p in proc]
    return ranks, min(process, n_ranks)
