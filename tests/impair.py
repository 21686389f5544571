"""Line impairments for the modem receivers' parity inputs (test infrastructure, numpy only).

The reference's own modulators put the carrier exactly on its nominal frequency and run on the receiver's sample clock, so a
receiver fed from them sits at the fixed point of its carrier loop (src/v29rx.c:297-331 track_carrier) and of its symbol
timing loop (src/godard.c:165-220).  A real line does neither: a carrier system shifts every frequency by a few hertz and
the far end's clock is some tens of ppm off.  These helpers apply both to an int16 signal; the receivers' expected outputs
for the result come from the oracle / the reference itself, as for any other input."""
import numpy as np


def frequency_shift(x, hz):
    """Single-sideband shift of every component of x by hz (what unsynchronised carrier-system oscillators do):
    Re(analytic(x) * exp(j 2 pi hz n / 8000))."""
    n = len(x)
    X = np.fft.fft(np.asarray(x, np.float64))
    h = np.zeros(n)
    h[0] = 1.0
    if n % 2 == 0:
        h[n//2] = 1.0
        h[1:n//2] = 2.0
    else:
        h[1:(n + 1)//2] = 2.0
    a = np.fft.ifft(X*h)
    return np.real(a*np.exp(2j*np.pi*hz*np.arange(n)/8000.0))


def resample_ppm(x, ppm, half=24):
    """x read with a sample clock ppm parts per million fast: output sample n is x(n*(1 + ppm*1e-6)), interpolated with a
    Hann-windowed sinc over 2*half input samples."""
    x = np.asarray(x, np.float64)
    n = len(x)
    t = np.arange(n)*(1.0 + ppm*1e-6)
    i0 = np.floor(t).astype(np.int64)
    frac = t - i0
    k = np.arange(-half + 1, half + 1)
    idx = i0[:, None] + k[None, :]
    d = k[None, :] - frac[:, None]
    w = np.sinc(d)*(0.5 + 0.5*np.cos(np.pi*d/half))
    ok = (idx >= 0) & (idx < n)
    return np.sum(np.where(ok, x[np.clip(idx, 0, n - 1)], 0.0)*w, axis=1)


def line(x, carrier_hz=0.0, ppm=0.0):
    """int16 in, int16 out: clock offset first (the far end's converter), then the frequency shift (the carrier system)."""
    y = np.asarray(x, np.float64)
    if ppm:
        y = resample_ppm(y, ppm)
    if carrier_hz:
        y = frequency_shift(y, carrier_hz)
    return np.clip(np.rint(y), -32768, 32767).astype(np.int16)
