"""Measured reference peaks of the box, to sit next to the nominal figures the roofline fractions use:
device-to-device copy and read-modify-write streams for HBM, a large bf16 product through the vendor library for the matrix cores."""
import torch


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3


def main():
    dev = "cuda"
    nbytes = 2 * 1024 ** 3
    a = torch.empty(nbytes // 4, dtype=torch.float32, device=dev).normal_()
    b = torch.empty_like(a)
    t = timed(lambda: b.copy_(a))
    print(f"copy  (read 2 GiB + write 2 GiB): {2 * nbytes / t / 1e12:.2f} TB/s")
    t = timed(lambda: a.mul_(1.0001))
    print(f"scale (read + write in place)    : {2 * nbytes / t / 1e12:.2f} TB/s")
    t = timed(lambda: a.sum())
    print(f"sum   (read only)                : {nbytes / t / 1e12:.2f} TB/s")
    for n in (4096, 8192):
        x = torch.randn(n, n, device=dev, dtype=torch.bfloat16)
        y = torch.randn(n, n, device=dev, dtype=torch.bfloat16)
        t = timed(lambda: torch.matmul(x, y), n=10)
        print(f"bf16 matmul {n}^3 (vendor library) : {2 * n ** 3 / t / 1e12:.0f} TFLOP/s")
    x = torch.randn(7968, 512, device=dev, dtype=torch.bfloat16)
    w = torch.randn(2048, 512, device=dev, dtype=torch.bfloat16)
    t = timed(lambda: torch.nn.functional.linear(x, w), n=50)
    print(f"bf16 7968x2048x512 (vendor library, plain) : {2 * 7968 * 2048 * 512 / t / 1e12:.0f} TFLOP/s, {t * 1e6:.1f} us")
    w2 = torch.randn(512, 2048, device=dev, dtype=torch.bfloat16)
    x2 = torch.randn(7968, 2048, device=dev, dtype=torch.bfloat16)
    t = timed(lambda: torch.nn.functional.linear(x2, w2), n=50)
    print(f"bf16 7968x512x2048 (vendor library, plain) : {2 * 7968 * 2048 * 512 / t / 1e12:.0f} TFLOP/s, {t * 1e6:.1f} us")


if __name__ == "__main__":
    main()
