"""Which shapes does torch's flash attention run fast on this box?  (the embedder's attention: 16 heads x 64, bf16, one segment of ~7.8 k tokens)"""
import time
import torch
import torch.nn.functional as F

def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n

H, D = 16, 64
def mk(T): return torch.randn(1, H, T, D, device="cuda", dtype=torch.bfloat16)
def tf(Tq, Tk, s): return 4 * Tq * Tk * H * D / s / 1e12
for Tq, Tk in ((7778, 7778), (7808, 7808), (7936, 7936), (8192, 8192), (7778, 7680), (7680, 7778), (7778, 7808), (7808, 7778), (7778, 128), (7778, 98), (7776, 7776), (7744, 7744)):
    q, k, v = mk(Tq), mk(Tk), mk(Tk)
    s = t(lambda: F.scaled_dot_product_attention(q, k, v))
    print(f"Tq {Tq} Tk {Tk}: {s*1e3:.3f} ms  {tf(Tq, Tk, s):.0f} TF")
T, Tp = 7778, 7936
q, k, v = mk(Tp), mk(Tp), mk(Tp)
madd = torch.zeros(1, 1, 1, Tp, device="cuda", dtype=torch.bfloat16); madd[..., T:] = float("-inf")
mbool = torch.ones(1, 1, 1, Tp, device="cuda", dtype=torch.bool); mbool[..., T:] = False
for name, m in (("additive mask", madd), ("bool mask", mbool)):
    s = t(lambda: F.scaled_dot_product_attention(q, k, v, attn_mask=m))
    print(f"padded {Tp} + {name}: {s*1e3:.3f} ms  {tf(T, T, s):.0f} TF (of the real work)")
# LSE merge: keys split into a multiple of 256 and the rest
try:
    q, k, v = mk(T), mk(T), mk(T)
    Tm = (T // 256) * 256
    def two():
        o1, l1 = torch.ops.aten._scaled_dot_product_flash_attention(q, k[:, :, :Tm], v[:, :, :Tm], 0.0, False, False)[:2]
        o2, l2 = torch.ops.aten._scaled_dot_product_flash_attention(q, k[:, :, Tm:], v[:, :, Tm:], 0.0, False, False)[:2]
        l = torch.logaddexp(l1, l2)
        return o1 * torch.exp(l1 - l).unsqueeze(-1).to(o1.dtype) + o2 * torch.exp(l2 - l).unsqueeze(-1).to(o2.dtype)
    ref = F.scaled_dot_product_attention(q, k, v)
    out = two()
    print("lse merge max err", float((out.float() - ref.float()).abs().max()), "lse shape", tuple(torch.ops.aten._scaled_dot_product_flash_attention(q, k, v, 0.0, False, False)[1].shape))
    s = t(two)
    print(f"two-call LSE merge: {s*1e3:.3f} ms  {tf(T, T, s):.0f} TF")
except Exception as e:
    print("lse merge:", type(e).__name__, str(e)[:300])
# varlen
try:
    q, k, v = (x.transpose(1, 2).reshape(T, H, D).contiguous() for x in (mk(T), mk(T), mk(T)))
    cu = torch.tensor([0, T], device="cuda", dtype=torch.int32)
    fn = lambda: torch.ops.aten._flash_attention_forward(q, k, v, cu, cu, T, T, 0.0, False, False)
    s = t(fn)
    print(f"varlen flash: {s*1e3:.3f} ms  {tf(T, T, s):.0f} TF")
except Exception as e:
    print("varlen:", type(e).__name__, str(e)[:300])
# fp8? / math
