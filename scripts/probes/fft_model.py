"""Index-math model of csrc/fft.hip's register-pass FFT (dev tool): passes A (radix NC/256), B (radix 16), C (radix 16),
LDS exchange layouts E1/E2/E3, checked against numpy.fft; also counts LDS bank conflicts per half-wave access."""
import numpy as np, sys

def model(LOGN, frames_in):
    NC = 1 << LOGN; T = NC // 16; FB = 256 // T; RA = NC // 256
    FS = NC + 256 + 2
    RS1 = T + 16; RS2 = T + 1; RS3 = 256 + (32 // RA if RA > 1 else 0)
    lds = np.zeros(FB * FS, dtype=np.complex128)
    W = lambda n, e: np.exp(-2j * np.pi * e / n)
    conf = {}
    def access(tag, addr_by_thread):      # addr_by_thread: (256,) addresses of one instruction
        worst = 1
        for hw in range(8):
            a = addr_by_thread[hw * 32:(hw + 1) * 32] % 32
            worst = max(worst, np.bincount(a, minlength=32).max())
        conf[tag] = max(conf.get(tag, 1), worst)
    tid = np.arange(256); fl = tid // T; u = tid % T
    z = np.zeros((256, 16), dtype=np.complex128)
    # global load: thread's points i = u + T*n
    for n in range(16):
        z[:, n] = frames_in[fl, u + T * n]
    if RA > 1:
        nb = 16 // RA                      # butterflies per thread; n = h + nb*q
        out = np.zeros_like(z)
        for h in range(nb):
            j1 = u + h * T                 # in [0, 256)
            x = z[:, h::nb]                # (256, RA) inputs q
            for m in range(RA):
                y = sum(x[:, q] * W(RA, q * m) for q in range(RA)) * W(NC, j1 * m)
                addr = fl * FS + (j1 // 16) * RS1 + m * 16 + (j1 % 16)
                access("A.write", addr); lds[addr] = y
        # pass B read
        for q in range(16):
            addr = fl * FS + q * RS1 + u
            access("B.read", addr); z[:, q] = lds[addr]
        b1 = u // 16; j2 = u % 16
    else:
        b1 = np.zeros(256, dtype=int); j2 = u
    y = np.zeros_like(z)
    for m in range(16):
        y[:, m] = sum(z[:, q] * W(16, q * m) for q in range(16)) * W(256, j2 * m)
    for m in range(16):
        addr = fl * FS + j2 * RS2 + b1 * 16 + m
        access("B.write", addr); lds[addr] = y[:, m]
    for q in range(16):
        addr = fl * FS + q * RS2 + u
        access("C.read", addr); z[:, q] = lds[addr]
    mA = u // 16; mB = u % 16
    def phys3(k):
        return (k % RA) * RS3 + k // RA
    for m in range(16):
        yv = sum(z[:, q] * W(16, q * m) for q in range(16))
        k = mA + RA * mB + 16 * RA * m
        addr = fl * FS + phys3(k)
        access("C.write", addr); lds[addr] = yv
    # split-step read pattern: lanes (fl fastest, k)
    idx = np.arange(256); f2 = idx % FB; k = idx // FB
    access("split.read", f2 * FS + phys3(k)); access("split.readm", f2 * FS + phys3((NC - k) % NC))
    Z = np.zeros((FB, NC), dtype=np.complex128)
    for f in range(FB):
        Z[f] = lds[f * FS + phys3(np.arange(NC))]
    return Z, conf

for LOGN in (8, 9, 10, 11):
    NC = 1 << LOGN; FB = 256 // (NC // 16)
    rng = np.random.default_rng(LOGN)
    x = rng.standard_normal((FB, NC)) + 1j * rng.standard_normal((FB, NC))
    Z, conf = model(LOGN, x)
    err = np.abs(Z - np.fft.fft(x, axis=1)).max()
    print(LOGN, "max err", err, conf)
