"""Stand-in for ``summertts_amd.engine`` used ONLY by tests/test_bench_launch_cpu.py: it lets bench.py's rank launcher,
sharding and gloo PCM gather run on a machine without a GPU.  It synthesises nothing -- ``run_batch`` returns a
deterministic ramp per utterance (100 samples per phoneme) -- and is never importable from the product package."""
import numpy as np

IS_STUB = True


class _Info:
    blob_floats_consumed = 0


class Synthesizer:
    def __init__(self, blob, device=0):
        self.info = _Info()
        self._pcm = np.zeros(0, np.int16)
        self._prof = {}

    def get_speaker_num(self):
        return 1

    def set_conv_mode(self, mode):
        pass

    def set_profiling(self, on):
        pass

    def run_batch(self, ids, sid=None, length_scale=None):
        n = np.asarray([len(a) * 100 for a in ids], np.int32)
        self._pcm = np.concatenate([(np.arange(k, dtype=np.int64) + int(a[0])).astype(np.int16) for a, k in zip(ids, n)])
        self._prof = {"frames": int(n.sum()) // 100, "samples": int(n.sum()), "phonemes": int(sum(len(a) for a in ids))}
        return n

    def pcm_host(self):
        return self._pcm

    def profile(self):
        return dict(self._prof)

    def close(self):
        pass


class MultiDevice:
    """Stand-in for engine.MultiDevice (`bench.py --multi native`): utterances dealt round-robin, the same ramp PCM as above."""
    _fake = None

    def __init__(self, blob, devices, gather="auto"):
        self.devices = list(devices)
        self.gather = gather
        self.h = None
        self.lib = None

    @staticmethod
    def set_rccl_library(path, allow_repeated_devices=False):
        MultiDevice._fake = path

    def gather_mode(self):
        return "rccl" if self.gather == "rccl" else "download"

    def rccl_ranks(self):
        return len(self.devices) if self.gather == "rccl" else 0

    def last_gather_ms(self):
        return 0.25

    def shard_of(self, lengths):
        return np.asarray([u % len(self.devices) for u in range(len(lengths))], np.int32)

    def infer_batch(self, ids, sid=None, length_scale=None):
        return [(np.arange(len(a) * 100, dtype=np.int64) + int(a[0])).astype(np.int16) for a in ids]

    def close(self):
        pass
