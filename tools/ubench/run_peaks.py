import ctypes, os
lib = ctypes.CDLL('/root/repo/tools/ubench/libsts_ubench.so')
for f in (lib.sts_ubench_mfma_bf16, lib.sts_ubench_mfma_f16):
    f.restype = ctypes.c_double; f.argtypes = [ctypes.c_int]*3
for blocks in (256, 512, 768):
    print(blocks, 'bf16 const %.0f split %.0f | f16 const %.0f two-term %.0f' % (lib.sts_ubench_mfma_bf16(0, blocks, 20000), lib.sts_ubench_mfma_bf16(2, blocks, 20000), lib.sts_ubench_mfma_f16(0, blocks, 40000), lib.sts_ubench_mfma_f16(2, blocks, 40000)), flush=True)
