"""Multi-GPU plumbing: one process per GPU, `torch.distributed` (backend "nccl" =
RCCL over xGMI on ROCm; "gloo" on CPU for tests).

The path shards by image tile (RtParams.tileRank / tileCount / tileSize): a
pixel's value depends only on its global coordinates, the per-frame seed and
the scene (Assets/Shaders/RayTracing/Header.cginc:79-92 keys the RNG on the
pixel), so ranks never exchange data while rendering.  Two exchanges remain:

  * broadcast_scene   - once per scene: rank 0's input arrays go to every rank,
                        which then builds its own (bit-identical) BVH;
  * TileGatherer      - once per finished image: every rank's accumulated tiles
                        go to rank 0, which places them in the full frame.
"""
import numpy as np
import torch
import torch.distributed as dist

from . import scenes, wire

SCENE_ARRAYS = ("vertices", "indices", "material_index", "shadow_index", "materials", "sphere_lights", "area_lights",
                "disc_lights", "skybox")
PAD = 0xFFFFFFFF


def broadcast_scene(sc, device, src=0):
    """Every rank returns the same scenes.Scene; only `src` needs to pass one."""
    if sc is not None and sc.primitives is not None:
        raise ValueError("broadcast_scene: the brute-force primitive lists are not sharded; trace them on one rank")
    rank = dist.get_rank()
    meta = [None]
    if rank == src:
        meta[0] = {k: (None if getattr(sc, k) is None else
                       (getattr(sc, k).dtype.descr if getattr(sc, k).dtype.names else getattr(sc, k).dtype.str,
                        getattr(sc, k).shape, getattr(sc, k).nbytes)) for k in SCENE_ARRAYS}
        meta[0]["_params"] = bytes(sc.params)
        meta[0]["_name"] = sc.name
    dist.broadcast_object_list(meta, src=src)
    m = meta[0]
    out = {}
    for k in SCENE_ARRAYS:
        if m[k] is None:
            out[k] = None
            continue
        descr, shape, nbytes = m[k]
        if rank == src:
            raw = np.frombuffer(np.ascontiguousarray(getattr(sc, k)).tobytes(), np.uint8).copy()
            t = torch.from_numpy(raw).to(device)
        else:
            t = torch.empty(nbytes, dtype=torch.uint8, device=device)
        if nbytes:
            dist.broadcast(t, src=src)
        dt = np.dtype([tuple(d) for d in descr]) if isinstance(descr, list) else np.dtype(descr)
        out[k] = np.frombuffer(t.cpu().numpy().tobytes(), dt).reshape(shape)
    if rank == src:
        return sc
    params = wire.RtParams.from_buffer_copy(m["_params"])
    return scenes.Scene(m["_name"], out["vertices"], out["indices"], out["material_index"], out["shadow_index"],
                        out["materials"], out["sphere_lights"], out["area_lights"], out["disc_lights"], out["skybox"], params)


def _group_turn(k, world):
    """By how many places the k-th group of `world` tiles is turned (the hash of
    RtTileMap::tile_of in csrc/rt_internal.h, in 64-bit integers)."""
    h = (np.asarray(k, dtype=np.int64) * 2654435761) & 0xFFFFFFFF
    h ^= h >> 16
    h = (h * 2246822507) & 0xFFFFFFFF
    h ^= h >> 13
    return (h >> 8) % world


def tile_owner(t, world):
    """Rank that renders tile t (row-major tile index; array or int): the
    inverse of RtTileMap::tile_of in csrc/rt_internal.h."""
    t = np.asarray(t, dtype=np.int64)
    turn = _group_turn(t // world, world)
    return (t % world - turn) % world


def tile_pixel_map(width, height, tile_size, rank, world):
    """Global pixel index (y*width+x, PAD for padding) of every tile-local slot of
    `rank` -- the host-side statement of rt_local_pixel_map: tiles of tile_size^2
    in row-major order, dealt to the ranks in groups of `world`, the k-th group
    turned by a hash of k (RtTileMap::tile_of in csrc/rt_internal.h); inside a
    tile 8x8 pixel blocks in row-major order, pixels row-major inside a block.
    Every rank has one tile of every group, so in the last group some ranks'
    tile may lie beyond the image: all padding."""
    tiles_x = (width + tile_size - 1) // tile_size
    tiles_y = (height + tile_size - 1) // tile_size
    total = tiles_x * tiles_y
    n_local = (total + world - 1) // world
    ts2 = tile_size * tile_size
    slot = np.arange(n_local * ts2, dtype=np.int64)
    k, within = slot // ts2, slot % ts2
    turn = _group_turn(k, world)
    t = k * world + (rank + turn) % world
    tx, ty = t % tiles_x, t // tiles_x
    b, lane = within >> 6, within & 63
    bpr = tile_size >> 3
    x = tx * tile_size + (b % bpr) * 8 + (lane & 7)
    y = ty * tile_size + (b // bpr) * 8 + (lane >> 3)
    ok = (x < width) & (y < height)
    return np.where(ok, y * width + x, PAD).astype(np.int64)


class TileGatherer:
    """Gathers every rank's tile-local float4 image on rank 0 and places it in
    the full frame.  Point-to-point fan-in (dist.gather): each peer has its own
    xGMI link to the root, so all transfers run concurrently; no ring.

    Every pixel belongs to exactly one rank, so the placement is a permutation
    known when the gatherer is made: `source[p]` is the slot of pixel p in the
    stacked receive buffer, and one index_select per image writes the frame in
    pixel order (no mask, no device-side count, no host synchronisation)."""

    def __init__(self, width, height, tile_size, device, dst=0, contexts=None):
        """`contexts` = n makes a gatherer for n contexts driven from THIS process
        (one per GPU, tileRank 0..n-1): no process group is touched and the images
        are handed to place() instead of gather()."""
        if contexts is not None:
            self.rank, self.world, dst = 0, int(contexts), 0
        else:
            self.rank, self.world = (dist.get_rank(), dist.get_world_size()) if dist.is_initialized() else (0, 1)
        self.in_process = contexts is not None
        self.dst, self.device = dst, device
        self.width, self.height, self.tile_size = width, height, tile_size
        maps = [tile_pixel_map(width, height, tile_size, r, self.world) for r in range(self.world)]
        self.n_local = len(maps[self.rank])
        self.max_local = max(len(m) for m in maps)
        self.send = torch.zeros(self.max_local * 4, dtype=torch.float32, device=device)
        self.stacked, self.recv, self.source, self.frame = None, None, None, None
        if self.rank == dst:
            source = np.full(width * height, -1, np.int64)
            for r, m in enumerate(maps):
                slots = np.nonzero(m != PAD)[0]
                source[m[slots]] = r * self.max_local + slots
            if (source < 0).any():
                raise AssertionError("tile maps do not cover the frame")
            self.source = torch.from_numpy(source).to(device)
            if self.world > 1:
                self.stacked = torch.zeros(self.world, self.max_local * 4, dtype=torch.float32, device=device)
                self.recv = list(self.stacked.unbind(0))        # contiguous rows: the gather list aliases `stacked`
            else:
                self.stacked = self.send.view(1, -1)
            self.frame = torch.zeros(width * height, 4, dtype=torch.float32, device=device)

    def local_buffer(self):
        """Flat float32 view (n_local*4) to be filled with this rank's tile-local image."""
        return self.send[: self.n_local * 4]

    def rank_buffer(self, rank):
        """In-process use: the flat float32 view context `rank` copies its tile-local image into."""
        n = len(tile_pixel_map(self.width, self.height, self.tile_size, rank, self.world))
        return self.stacked[rank, : n * 4]

    def place(self):
        """In-process use: the (height, width, 4) frame made of what the contexts wrote into their rank_buffer()s."""
        torch.index_select(self.stacked.view(-1, 4), 0, self.source, out=self.frame)
        return self.frame.view(self.height, self.width, 4)

    def gather(self):
        """Collective.  Returns the (height, width, 4) frame on the destination rank, None elsewhere."""
        if self.in_process:
            raise RuntimeError("this gatherer serves contexts of one process: use rank_buffer() and place()")
        if self.world > 1:
            dist.gather(self.send, self.recv if self.rank == self.dst else None, dst=self.dst)
        if self.rank != self.dst:
            return None
        torch.index_select(self.stacked.view(-1, 4), 0, self.source, out=self.frame)
        return self.frame.view(self.height, self.width, 4)
