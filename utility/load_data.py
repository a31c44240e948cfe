"""Dataset container with the reference's attributes and sampling semantics
(reference utility/load_data.py:10-92,157-195).

``sample()`` draws from python's ``random`` and ``np.random`` global streams in exactly the
reference's call order, so the same seed yields the same (users, pos, neg) triples as the
reference (tests/test_oracle_golden.py::test_host_sampler_stream_matches_reference).
``sample_device()`` is the MI355X-native alternative: Philox-keyed sampling in one HIP launch
(llmrec_sample_bpr), distribution-equivalent but not stream-equivalent."""
import json
import random as rd

import numpy as np

from utility.parser import parse_args

args = parse_args()


class Data(object):
    def __init__(self, path, batch_size):
        self.path = path
        self.batch_size = batch_size
        with open(path + '/train.json') as f:
            train = json.load(f)
        with open(path + '/test.json') as f:
            test = json.load(f)
        with open(path + '/val.json') as f:
            val = json.load(f)

        self.neg_pools = {}
        self.exist_users = []
        self.train_items, self.test_set, self.val_set = {}, {}, {}
        self.n_train = self.n_test = self.n_val = 0
        max_uid = 0
        for uid, items in train.items():
            if len(items) == 0:
                continue
            uid = int(uid)
            self.exist_users.append(uid)
            max_uid = max(max_uid, uid)
            self.n_train += len(items)
            self.train_items[uid] = items
        for uid, items in test.items():
            if len(items):
                self.test_set[int(uid)] = items
                self.n_test += len(items)
        for uid, items in val.items():
            if len(items):
                self.val_set[int(uid)] = items
                self.n_val += len(items)
        self.n_users = max_uid + 1                                   # from train.json only, as the reference
        # the item count is defined by the text-feature matrix (reference load_data.py:57-58)
        self.n_items = np.load(args.data_path + args.dataset + '/text_feat.npy', mmap_mode='r').shape[0]
        self._train_sets = {u: set(v) for u, v in self.train_items.items()}
        self._R = None
        self._fast_sampler = None                                    # decided on the first batch (sample())
        self._host = None                                            # the C helper of the draw loop, loaded on first use
        self._fast_users, self._users_scratch, self._exist_arr = {}, None, None   # the C replay of random.sample: verified once PER BRANCH (pool / set) of CPython's algorithm; False = off
        self._device_state = None
        self.print_statistics()

    @property
    def R(self):
        """User x item interaction matrix (scipy); built on first use (the reference fills a DOK
        matrix element by element at import, load_data.py:63-72, and then never reads it)."""
        if self._R is None:
            import scipy.sparse as sp
            rows = np.concatenate([np.full(len(v), u, dtype=np.int64) for u, v in self.train_items.items()])
            cols = np.concatenate([np.asarray(v, dtype=np.int64) for v in self.train_items.values()])
            self._R = sp.csr_matrix((np.ones(rows.size, dtype=np.float32), (rows, cols)),
                                    shape=(self.n_users, self.n_items)).todok()
        return self._R

    def sample(self):
        if self.batch_size <= self.n_users:
            users = self._sample_users()
        else:
            users = [rd.choice(self.exist_users) for _ in range(self.batch_size)]
        if self._fast_sampler is None:                               # first batch: both forms from the same state must agree
            state = np.random.get_state()
            slow = self._draw_items_reference(users)
            after = np.random.get_state()
            np.random.set_state(state)
            try:
                fast = self._draw_items_fast(users)
                now = np.random.get_state()
                self._fast_sampler = bool(fast == slow and now[2] == after[2] and np.array_equal(now[1], after[1]))
            except Exception:
                self._fast_sampler = False
            if not self._fast_sampler:                               # an unknown numpy draws differently: keep the per-call form
                print("utility.load_data: the block form of Data.sample() does not reproduce np.random.randint's stream "
                      "with numpy %s; using the per-call form" % np.__version__)
            np.random.set_state(after)
            return users, slow[0], slow[1]
        pos_items, neg_items = self._draw_items_fast(users) if self._fast_sampler else self._draw_items_reference(users)
        return users, pos_items, neg_items

    def _sample_users(self):
        """rd.sample(self.exist_users, self.batch_size) (reference load_data.py:159)."""
        if self._exist_arr is None:
            self._exist_arr = np.asarray(self.exist_users, dtype=np.int64)
        return self.py_sample(self.exist_users, self.batch_size, self._exist_arr)

    def py_sample(self, population, k, population_arr=None):
        """random.sample(population, k) on python's global generator - through the C replay of CPython's random.sample on CPython's own
        generator state when the host helper is there (llmrec_host_py_sample: 0.19 -> 0.07 ms at k = 1024 of 13 187); the first call draws
        both ways from the same state and compares, as the item draws do - once for each of random.sample's two branches (the pool branch, n <= setsize:
        the per-step augmented-triple draw; the set branch: the exist_users draw): ADVICE r04. population_arr: the population as an int64 array, if the caller has it."""
        host = self._host_lib()
        n = len(population)
        if self._fast_users is None:                                 # (tests reset the latch with None)
            self._fast_users = {}
        if host is None or self._fast_users is False or k > n or k <= 0:
            return rd.sample(population, k)
        import math
        setsize = 21
        if k > 5:
            setsize += 4 ** math.ceil(math.log(k * 3, 4))           # (random.sample's own expression, Lib/random.py)
        st = rd.getstate()
        if st[0] != 3 or len(st[1]) != 625:
            self._fast_users = False
            return rd.sample(population, k)
        words = np.array(st[1], dtype=np.uint32)
        need = max(n, (n + 63) // 64)
        if self._users_scratch is None or self._users_scratch.size < need:
            self._users_scratch = np.empty(need, dtype=np.int64)
        pos = np.empty(k, dtype=np.int64)
        use_pool = n <= setsize
        if self._fast_users.get(use_pool) is False:
            return rd.sample(population, k)
        if host.llmrec_host_py_sample(words.ctypes.data, n, k, 1 if use_pool else 0, self._users_scratch.ctypes.data, pos.ctypes.data) != 0:
            self._fast_users = False
            return rd.sample(population, k)
        arr = population_arr if population_arr is not None else np.asarray(population, dtype=np.int64)
        out = arr[pos].tolist()
        if use_pool not in self._fast_users:                         # first call of this branch: the interpreter's own random.sample must agree
            want = rd.sample(population, k)                          # (advances the stream exactly as the replay claims to have done)
            after = rd.getstate()
            ok = self._fast_users[use_pool] = bool(want == out and tuple(words.tolist()) == after[1])
            if not ok:
                print("utility.load_data: the C replay of random.sample (%s branch) does not reproduce this interpreter's stream; using random.sample"
                      % ("pool" if use_pool else "set"))
            return want
        rd.setstate((st[0], tuple(words.tolist()), st[2]))
        return out

    def _draw_items_reference(self, users):
        """One positive and one rejected negative per user, one np.random.randint(size=1) call per draw as the reference
        makes them (load_data.py:163-186)."""
        pos_items, neg_items = [], []
        for u in users:
            mine = self.train_items[u]
            pos_items.append(mine[np.random.randint(low=0, high=len(mine), size=1)[0]])
            seen = self._train_sets[u]
            while True:
                neg_id = np.random.randint(low=0, high=self.n_items, size=1)[0]
                if neg_id not in seen:
                    neg_items.append(neg_id)
                    break
        return pos_items, neg_items

    def _draw_items_fast(self, users):
        """The same draws from the same global np.random stream, without ~2.1 k scalar randint calls per batch (7 ms at B = 1024,
        14x the GPU step). RandomState.randint(0, high, size=1) with the default int64 dtype is masked rejection over raw 32-bit
        words of the MT19937 stream: rng = high - 1; no word at all when rng == 0; else words & mask (mask = the next 2^k - 1 >= rng)
        until one is <= rng. So: fetch a block of raw words (randint over the full uint32 range returns them one per value), replay
        the reference's draw sequence on it in plain integers, then advance the global stream by exactly the words consumed. The
        first batch of a run is drawn both ways and compared (sample())."""
        state = np.random.get_state()
        host = self._host_lib()
        if host is not None:                                         # the same replay in C (llmrec_amd/csrc/host_sampler.c)
            import ctypes
            u = np.asarray(users, dtype=np.int64)
            pos, neg = np.empty(u.size, dtype=np.int64), np.empty(u.size, dtype=np.int64)
            need = 4 * u.size + 64
            while True:
                raw = np.random.randint(0, 1 << 32, size=need, dtype=np.uint32)
                k = host.llmrec_host_draw_items(u.size, u.ctypes.data, self._list_ptr.ctypes.data, self._list_items.ctypes.data, self.n_items,
                                                raw.ctypes.data, raw.size, pos.ctypes.data, neg.ctypes.data)
                np.random.set_state(state)
                if k >= 0:
                    break
                if k == -2:
                    raise RuntimeError("Data.sample: a user owns every item - no negative exists")
                need *= 2
            if k:
                np.random.randint(0, 1 << 32, size=int(k), dtype=np.uint32)
            return pos.tolist(), neg.tolist()
        n_mask = self._mask(self.n_items - 1)
        n_rng = self.n_items - 1
        need = 4 * len(users) + 64
        while True:
            raw = np.random.randint(0, 1 << 32, size=need, dtype=np.uint32).tolist()
            k = 0
            pos_items, neg_items = [], []
            try:
                for u in users:
                    mine = self.train_items[u]
                    rng = len(mine) - 1
                    if rng == 0:
                        pos_items.append(mine[0])
                    else:
                        m = self._mask(rng)
                        while True:
                            v = raw[k] & m
                            k += 1
                            if v <= rng:
                                break
                        pos_items.append(mine[v])
                    seen = self._train_sets[u]
                    while True:
                        if n_rng == 0:
                            v = 0
                        else:
                            while True:
                                v = raw[k] & n_mask
                                k += 1
                                if v <= n_rng:
                                    break
                        if v not in seen:
                            neg_items.append(v)
                            break
                break
            except IndexError:                                       # block too short (many rejections): restart with a longer one
                np.random.set_state(state)
                need *= 2
        np.random.set_state(state)
        if k:
            np.random.randint(0, 1 << 32, size=k, dtype=np.uint32)   # the stream position the per-call form would have left
        return pos_items, neg_items

    def _host_lib(self):
        """libllmrec_host.so (gcc, llmrec_amd/build.py::build_host) + the train lists as one int64 array with offsets, or None."""
        if self._host is None:
            self._host = False
            try:
                import ctypes
                import os
                so = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "llmrec_amd", "lib", "libllmrec_host.so")
                if os.path.exists(so):
                    lib = ctypes.CDLL(so)
                    lib.llmrec_host_draw_items.restype = ctypes.c_int64
                    lib.llmrec_host_draw_items.argtypes = [ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64,
                                                           ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p]
                    lib.llmrec_host_py_sample.restype = ctypes.c_int32
                    lib.llmrec_host_py_sample.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p]
                    ptr = np.zeros(self.n_users + 1, dtype=np.int64)
                    for u_, items in self.train_items.items():
                        ptr[u_ + 1] = len(items)
                    np.cumsum(ptr, out=ptr)
                    flat = np.zeros(int(ptr[-1]), dtype=np.int64)
                    for u_, items in self.train_items.items():
                        flat[ptr[u_]:ptr[u_ + 1]] = items                # the file's order: positives are drawn by position
                    self._list_ptr, self._list_items, self._host = ptr, flat, lib
            except Exception:
                self._host = False
        return self._host or None

    @staticmethod
    def _mask(rng):
        m = rng
        for sh in (1, 2, 4, 8, 16):
            m |= m >> sh
        return m

    # -- device-side state (train CSR etc.), shared by the sampler and the evaluator ------------
    def device_state(self, device):
        import torch
        from llmrec_amd import ops
        device = torch.device(device)
        if device.type == "cuda" and device.index is None:           # "cuda" and "cuda:<current>" are the same device: one cached state,
            device = torch.device("cuda", torch.cuda.current_device())   # not one rebuilt (55 ms at Netflix shape) at every other call
        if self._device_state is not None and self._device_state["device"] == device:
            return self._device_state

        def csr_of(d):
            if not d:
                z = torch.zeros(self.n_users + 1, dtype=torch.int32, device=device)
                return z, torch.zeros(0, dtype=torch.int32, device=device)
            rows = np.concatenate([np.full(len(v), u, dtype=np.int64) for u, v in d.items()])
            cols = np.concatenate([np.asarray(v, dtype=np.int64) for v in d.values()])
            rp, ci, _ = ops.csr_from_coo(torch.from_numpy(rows).to(device), torch.from_numpy(cols).to(device), None,
                                         self.n_users, self.n_items)
            return rp, ci
        tr = csr_of(self.train_items)
        train = ops.Csr(self.n_users, self.n_items, tr[0], tr[1], None, None, None, ops.SpmmPlan())
        self._device_state = {"device": device, "train": train, "test": csr_of(self.test_set), "val": csr_of(self.val_set),
                              "exist_users": torch.tensor(self.exist_users, dtype=torch.int64, device=device)}
        return self._device_state

    def sample_device(self, seed, step, device):
        """(users, pos, neg) int64 device tensors from the HIP sampler."""
        from llmrec_amd import ops
        st = self.device_state(device)
        return ops.sample_bpr(seed, step, st["exist_users"], self.n_items, st["train"], self.batch_size)

    def print_statistics(self):
        print('n_users=%d, n_items=%d' % (self.n_users, self.n_items))
        print('n_interactions=%d' % (self.n_train + self.n_test))
        print('n_train=%d, n_test=%d, sparsity=%.5f' % (self.n_train, self.n_test,
                                                        (self.n_train + self.n_test) / (self.n_users * self.n_items)))
