"""lone 2^16 verifications through bgls_miller_product_dev: with / without the duplicate scan, in throughput mode (other hashing schedule)"""
import ctypes, os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from bgls_amd import _lib
import bench
L = _lib.load(); assert L.bgls_init(0) == 0
dev = torch.device("cuda", 0)
cid, n = 0, 65536
inst = bench.make_instance(L, cid, n, 5)
fp = inst["fp"]
t_keys = torch.frombuffer(bytearray(inst["keys"]), dtype=torch.uint8).to(dev)
t_msgs = torch.frombuffer(bytearray(inst["msgs"]), dtype=torch.uint8).to(dev)
t_sig = torch.frombuffer(bytearray(bench.aggregate_sig(L, inst, 0, n)), dtype=torch.uint8).to(dev)
part = torch.zeros(12 * fp, dtype=torch.uint8, device=dev); flags = torch.zeros(1, dtype=torch.int32, device=dev)
mode = sys.argv[1]
if mode == "tp": L.bgls_set_throughput_mode(1)
st = torch.cuda.current_stream().cuda_stream
for _ in range(24):
    flags.zero_()
    assert L.bgls_miller_product_dev(cid, t_sig.data_ptr(), t_keys.data_ptr(), t_msgs.data_ptr(), 64, 64, n, 0 if mode == "nodup" else 1, part.data_ptr(), flags.data_ptr(), st) >= 0
    assert L.bgls_final_verify_dev(cid, part.data_ptr(), 1, flags.data_ptr(), st) == 1
print("done")
