"""Exercise, over the real nccl (RCCL) backend with world_size 1, every collective and dtype bench.py uses at N > 1."""
import os
import torch
import torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
for dtype, shape in ((torch.float32, (64, 1024, 3)), (torch.uint8, (64, 1024, 4))):
    t = (torch.rand(shape, device=dev) * 200).to(dtype)
    out = [torch.empty_like(t)]
    dist.gather(t, gather_list=out, dst=0)
    assert torch.equal(out[0], t), dtype
# FilmGather's form: uint8 [rows, W, 16] tile gathered into views of one contiguous receive buffer
send = (torch.rand((128, 1024, 16), device=dev) * 255).to(torch.uint8)
recv = torch.empty((1, 128, 1024, 16), dtype=torch.uint8, device=dev)
dist.gather(send, gather_list=[recv[0]], dst=0)
assert torch.equal(recv[0], send)
# ... launched asynchronously and completed later (FilmGather.start / finish), twice in a row on the same buffers
for k in range(2):
    send.add_(1)
    work = dist.gather(send, gather_list=[recv[0]], dst=0, async_op=True)
    busy = torch.rand((2048, 2048), device=dev) @ torch.rand((2048, 2048), device=dev)     # other work on the current stream
    work.wait()
    assert torch.equal(recv[0], send), k
x = torch.tensor([1.5], dtype=torch.float64, device=dev); dist.all_reduce(x, op=dist.ReduceOp.MAX); assert x.item() == 1.5
y = torch.tensor([3.0, 4.0], dtype=torch.float64, device=dev); dist.all_reduce(y, op=dist.ReduceOp.SUM); assert y.tolist() == [3.0, 4.0]
dist.barrier()
torch.cuda.synchronize()
dist.destroy_process_group()
print("nccl smoke ok")
