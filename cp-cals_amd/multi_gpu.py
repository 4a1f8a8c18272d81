"""CALS over the GPUs of one node with a pull-based work queue.

Models are independent given X, so every rank (= one GPU) runs its own engine on a replica of X and
no factor data ever crosses GPUs during a run.  What differs from the static round-robin shards of
`sharding.py` is the ASSIGNMENT: with tolerance-driven convergence some models leave after 5 sweeps
and others after 200, so static shards drain unevenly.  Here the model indices sit behind one shared
counter; whenever a rank's engine has room in its column buffer it claims the next few indices,
materialises those models (only the claiming rank ever touches a model's data) and enqueues them,
while its engine keeps sweeping (`cals_hip_step`).  The hand-off is index-sized: one atomic add on the
process group's key-value store per claim -- no collective on the data path, and none is invented.
Results stay on the rank that fitted them; `gather_results` brings them together over a gloo group
(host objects; RCCL moves device tensors only).
"""
import os
import time

import torch.distributed as dist


class WorkQueue:
    """Indices 0 .. n_items-1 handed out in claim order through an atomic counter on a TCPStore."""

    # default-store queues are constructed collectively (every rank, same order): the n-th one everywhere.  Only those
    # count -- a queue with an explicit store, a single-process queue or one built before the process group exists
    # must not shift the key prefix of the ranks that happened to build one.
    _instances = 0

    def __init__(self, n_items, name="cals_work_queue", store=None, port=None):
        """store: any torch.distributed store with add(); default = the process group's own rendezvous
        store behind a prefix (no extra port).  port: only if that store is not reachable -- a TCPStore on
        MASTER_ADDR:port, chosen by the caller (never guessed from MASTER_PORT).  Construct queues in the same
        order on every rank: each instance counts under its own key prefix, so a second queue of the same name
        in one job starts at 0 again instead of at the first one's final value."""
        self.n = int(n_items)
        self.key = name
        self._local = 0  # single-process fall-back
        self.store = store
        self._members_ok = True  # default-store queues: checked at the first claim (see _check_members)
        if self.store is None and dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            if port is None:
                WorkQueue._instances += 1
                get_store = getattr(dist.distributed_c10d, "_get_default_store", None)
                if get_store is None:
                    raise RuntimeError("this torch build does not expose the process group's store "
                                       "(distributed_c10d._get_default_store): pass store= or port= to WorkQueue")
                base = get_store()
                self.store = dist.PrefixStore("cals_work_queue/%d/%s" % (WorkQueue._instances, name), base)
                self.store.add(self.key + "/members", 1)
                self._members_ok = False
            else:
                host = os.environ.get("MASTER_ADDR", "127.0.0.1")
                self.store = dist.TCPStore(host, int(port), dist.get_world_size(), is_master=(dist.get_rank() == 0),
                                           wait_for_workers=True)

    def _check_members(self, timeout=120.0):
        """Every rank of the group must count under THIS key: ranks that built a different number of queues would
        each hand out every index (the work silently done twice).  The first claim waits until all ranks have
        registered under the prefix and raises if they never do."""
        world = dist.get_world_size()
        t_end = time.time() + timeout
        while True:
            seen = self.store.add(self.key + "/members", 0)
            if seen == world:
                break
            if seen > world or time.time() > t_end:
                raise RuntimeError("WorkQueue %r: %d of %d ranks registered under this key prefix -- the ranks did not "
                                   "construct their work queues in the same order" % (self.key, seen, world))
            time.sleep(0.01)
        self._members_ok = True

    def claim(self, count):
        """Claims up to `count` indices; returns a (possibly empty) list."""
        count = int(count)
        if count <= 0:
            return []
        if not self._members_ok:
            self._check_members()
        if self.store is None:
            lo = self._local
            self._local += count
        else:
            lo = self.store.add(self.key, count) - count  # add() returns the value after the addition
        hi = min(lo + count, self.n)
        return list(range(lo, hi)) if lo < self.n else []


def cp_cals_work_queue(engine, n_models, make_model, queue=None, claim_models=8, on_done=None):
    """Runs CALS on `engine` (tensor and params already set) over the models that this rank claims from
    `queue`, until the queue is empty and the engine has drained.

    make_model(k) -> cp_cals_amd.Model for global index k (called only by the claiming rank).
    A rank claims `claim_models` more indices whenever none of its claimed models is waiting for
    buffer columns any more -- a rank whose buffer is full leaves the rest to the others.
    Returns ({k: model}, stats)."""
    if queue is None:
        queue = WorkQueue(n_models)
    mine, drained, claims, sweeps = {}, False, 0, 0
    t0 = time.time()
    while True:
        if not drained and engine.queue_size == 0:
            ks = queue.claim(claim_models)
            claims += 1
            if not ks:
                drained = True
            for k in ks:
                m = make_model(k)
                mine[k] = m
                engine.enqueue(m)
        if engine.queue_size == 0 and engine.models_in_flight == 0:
            if drained:
                break
            continue
        _, evicted = engine.step()
        sweeps += 1
        if evicted and on_done is not None:
            on_done(evicted)
    for m in mine.values():
        engine.result(m)
    rep = engine.report()
    stats = {"models": len(mine), "sweeps": sweeps, "claims": claims, "seconds": time.time() - t0,
             "ls_performed": rep.ls_performed, "ls_failed": rep.ls_failed}
    return mine, stats


def gather_results(mine, group=None):
    """{k: (factors, lambda, iters, error, fit)} of all ranks on every rank (gloo group for host objects)."""
    payload = {k: ([f.copy() for f in m.factors], m.lam.copy(), m.iters, m.error, m.fit) for k, m in mine.items()}
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return payload
    if group is None and dist.get_backend() != "gloo":
        group = dist.new_group(backend="gloo")
    parts = [None] * dist.get_world_size()
    dist.all_gather_object(parts, payload, group=group)
    out = {}
    for p in parts:
        out.update(p)
    return out
