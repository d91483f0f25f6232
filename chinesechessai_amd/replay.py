"""Host mirror of trainer.py's ReplayBuffer (trainer.py:22-44) on a device-resident ring.

    buf = ReplayBuffer(max_size=10000)
    buf.push_records(records_tensor, n_games)     # engine.pack_samples() / all-gathered shards
    buf.push(game_data)                           # reference-format [(board, {move: p}, z)] tuples
    states, targets = buf.sample_tensors(64)      # float32 [64,15,10,9], [64,1] on the GPU
    boards, move_probs_list, rewards = buf.sample(64)   # the reference's return value

`sample*` draws its indices exactly like the reference: np.random.choice(len, batch, replace=False)
on NumPy's global stream (trainer.py:37).  `sample` hands back the very tuples `push` was given
(trainer.py:27-42 keeps the tuples themselves): a host-side table beside the device ring remembers
them per ring position, so pushed probabilities come back exactly; records that arrived from the
device (`push_records`) are decoded from their visit counts.  States are encode_board(board, 1) — the player plane is
hard-wired to red in the reference's trainer (trainer.py:314) — and only z is used as a target
(value-only loss, trainer.py:324-331).
"""
import collections
import ctypes as C

import numpy as np

from . import _lib
from .chess_env import decode_move, encode_move
from .distributed import RECORD_BYTES, RECORD_DTYPE, record_to_sample


def pack_board_words(board):
    b = np.asarray(board, dtype=np.int16).reshape(90)
    code = np.where(b > 0, b, np.where(b < 0, 7 - b, 0)).astype(np.uint32)
    w = np.zeros(12, np.uint32)
    for s in range(90):
        w[s // 8] |= code[s] << np.uint32(4 * (s % 8))
    return w


class ReplayBuffer:
    def __init__(self, max_size=10000, device=0, temperature=1.0):
        self.L = _lib.lib()
        h = C.c_void_p()
        rc = self.L.xq_replay_create(device, max_size, C.byref(h))
        if rc != 0:
            raise _lib.XqError("xq_replay_create failed (%d): %s — the replay buffer is device-resident, "
                               "there is no CPU fallback" % (rc, self.L.xq_replay_last_error().decode()))
        self.h = h
        self.max_size = max_size
        self.temperature = temperature
        # one entry per ring position, oldest first, in step with the device ring (same maxlen, same append order):
        # the tuple `push` was given, or None for a record that came from the device
        self._host = collections.deque(maxlen=max_size)

    def _chk(self, rc):
        if rc != 0:
            raise _lib.XqError("libxq_hip replay error %d: %s" % (rc, self.L.xq_replay_last_error().decode()))

    def close(self):
        if getattr(self, "h", None):
            self.L.xq_replay_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __len__(self):
        return int(self.L.xq_replay_size(self.h))

    @staticmethod
    def _stream():
        import torch
        return torch.cuda.current_stream().cuda_stream

    def push_records(self, records, n_games):
        """records: uint8 CUDA tensor holding xq_sample_record[n_games][70] (engine.pack_samples or
        distributed.all_gather_records output).  Appends every valid record, oldest dropped first."""
        assert records.is_cuda and records.numel() >= n_games * _lib.MAX_PLIES * RECORD_BYTES
        n = C.c_int64()
        self._chk(self.L.xq_replay_push_records(self.h, C.c_void_p(self._stream()), C.c_void_p(records.data_ptr()),
                                                 int(n_games), C.byref(n)))
        self._host.extend([None] * min(n.value, self.max_size))
        return n.value

    def push(self, game_data):
        """trainer.py:27-33 for one game in the reference's tuple format [(board, {move: prob}, z)].  Boards and z
        go to the device ring (they are what batch formation reads, trainer.py:313-321); the tuples themselves are
        kept host-side and returned by sample() unchanged, as the reference's deque does.  Nothing is truncated: a
        game longer than 70 samples spans several 70-record blocks; a sample with more than 128 moves does not fit
        the record and raises."""
        import torch
        game_data = list(game_data)
        if not game_data:
            return 0
        nblk = (len(game_data) + _lib.MAX_PLIES - 1) // _lib.MAX_PLIES
        rec = np.zeros((nblk, _lib.MAX_PLIES), dtype=RECORD_DTYPE)
        for i, (board, move_probs, z) in enumerate(game_data):
            r = rec[i // _lib.MAX_PLIES, i % _lib.MAX_PLIES]
            moves = list(move_probs.keys())
            if len(moves) > _lib.MAX_MOVES:
                raise ValueError("ReplayBuffer.push: sample %d has %d moves; a sample record holds %d"
                                 % (i, len(moves), _lib.MAX_MOVES))
            r["board"] = pack_board_words(board)
            r["z"] = float(z)
            r["player"] = 0
            r["n_moves"] = len(moves)
            r["valid"] = 1
            r["moves"][:len(moves)] = [encode_move(m) for m in moves]
            # (the record's 16-bit counts cannot hold probabilities: they stay a coarse view for tools that read raw
            # records; sample() returns the pushed tuple itself)
            r["counts"][:len(moves)] = [int(round(min(max(float(move_probs[m]), 0.0), 1.0) * 65535)) for m in moves]
        t = torch.from_numpy(np.frombuffer(rec.tobytes(), dtype=np.uint8).copy()).cuda()
        n = C.c_int64()
        self._chk(self.L.xq_replay_push_records(self.h, C.c_void_p(self._stream()), C.c_void_p(t.data_ptr()),
                                                 int(nblk), C.byref(n)))
        assert n.value == len(game_data)
        self._host.extend(game_data)
        return n.value

    def _indices(self, batch_size):
        return np.ascontiguousarray(np.random.choice(len(self), batch_size, replace=False), dtype=np.int64)   # trainer.py:37

    def sample_tensors(self, batch_size, indices=None):
        """Device-side batch formation: (states float32 [B,15,10,9], targets float32 [B,1])."""
        import torch
        idx = self._indices(batch_size) if indices is None else np.ascontiguousarray(indices, dtype=np.int64)
        states = torch.empty((len(idx), 15, 10, 9), dtype=torch.float32, device="cuda")
        targets = torch.empty((len(idx), 1), dtype=torch.float32, device="cuda")
        self._chk(self.L.xq_replay_encode_batch(self.h, C.c_void_p(self._stream()), _lib.ptr(idx), len(idx),
                                                 C.c_void_p(states.data_ptr()), C.c_void_p(targets.data_ptr())))
        return states, targets

    def sample(self, batch_size, indices=None):
        """trainer.py:35-42: (boards, move_probs_list, rewards) tuples of length batch_size."""
        idx = self._indices(batch_size) if indices is None else np.ascontiguousarray(indices, dtype=np.int64)
        assert len(self._host) == len(self)
        rec = np.zeros(len(idx), dtype=RECORD_DTYPE)
        self._chk(self.L.xq_replay_read_records(self.h, C.c_void_p(self._stream()), _lib.ptr(idx), len(idx), _lib.ptr(rec)))
        out = [self._host[int(i)] if self._host[int(i)] is not None else record_to_sample(r, self.temperature)
               for i, r in zip(idx, rec)]
        boards, probs, rewards = zip(*out)
        return boards, probs, rewards
