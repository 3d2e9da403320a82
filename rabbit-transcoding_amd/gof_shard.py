"""GOF sharding across the GPUs of a node and the gather of the re-encoded NAL units.

A V3C group of frames is self-contained (it starts with a VPS, PCCBitstreamReader.cpp:78-96, and every video
sub-bitstream restarts with an IDR), so GOFs shard across ranks with no data-path collective; the only exchange is
the gather of the re-encoded sub-bitstreams (<= ~2 MB per GOF at R3) onto rank 0, which writes the output file
(PccAppTranscoder.cpp:345-348). One process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI on ROCm,
"gloo" in the CPU tests).
"""
from typing import List, Sequence


def gofs_of_rank(n_gofs: int, rank: int, world: int) -> List[int]:
    """GOF g is processed by rank g mod world (round-robin keeps ranks balanced when n_gofs is not a multiple)."""
    return [g for g in range(n_gofs) if g % world == rank]


def gather_streams(local: Sequence[bytes], group=None, device="cpu") -> List[List[bytes]]:
    """All ranks call this with their list of byte strings (one per locally transcoded sub-bitstream, in GOF order).
    Returns on rank 0 a list indexed by rank of those lists; on other ranks an empty list.
    Two collectives per call: an all_gather of the sizes, then an all_gather of the padded payloads (RCCL has no
    gatherv; the payloads are a few MB, far below what xGMI moves in a millisecond)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    sizes = torch.tensor([len(b) for b in local], dtype=torch.int64, device=device)
    n_local = torch.tensor([len(local)], dtype=torch.int64, device=device)
    counts = [torch.zeros(1, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(counts, n_local, group=group)
    max_n = int(max(int(c.item()) for c in counts))
    sizes_p = torch.zeros(max_n, dtype=torch.int64, device=device)
    sizes_p[: len(local)] = sizes
    all_sizes = [torch.zeros(max_n, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(all_sizes, sizes_p, group=group)
    totals = [int(s.sum().item()) for s in all_sizes]
    cap = max(1, max(totals))
    payload = torch.zeros(cap, dtype=torch.uint8, device=device)
    blob = b"".join(local)
    if blob:
        payload[: len(blob)] = torch.frombuffer(bytearray(blob), dtype=torch.uint8).to(device)
    gathered = [torch.zeros(cap, dtype=torch.uint8, device=device) for _ in range(world)]
    dist.all_gather(gathered, payload, group=group)
    if rank != 0:
        return []
    out = []
    for r in range(world):
        raw = gathered[r][: totals[r]].cpu().numpy().tobytes()
        n = int(counts[r].item())
        items, pos = [], 0
        for k in range(n):
            sz = int(all_sizes[r][k].item())
            items.append(raw[pos:pos + sz])
            pos += sz
        out.append(items)
    return out


def stitch(gathered: List[List[bytes]], n_gofs: int, streams_per_gof: int) -> List[List[bytes]]:
    """Re-orders what gather_streams returned into GOF order: result[g][s] = sub-bitstream s of GOF g."""
    world = len(gathered)
    res = [[b""] * streams_per_gof for _ in range(n_gofs)]
    for r in range(world):
        mine = gofs_of_rank(n_gofs, r, world)
        for i, g in enumerate(mine):
            for s in range(streams_per_gof):
                res[g][s] = gathered[r][i * streams_per_gof + s]
    return res
