"""How many Gauss-Legendre nodes along the segment make  sum_n w_n gT gT gL  equal to the closed form of detsim.py:120-159
(2 erf + exp) to 1e-12 of the peak density, as a function of (segment length) / (Gaussian width along the segment).
This is where qweights_kernel's rule  N = ceil(6 + 1.9 * ratio)  comes from (kernels_qweights.hip).  CPU only (numpy/scipy)."""
import numpy as np
from numpy.polynomial.legendre import leggauss
from scipy.special import erf


def closed(x, y, z, s, D, sT, sL, q=1.0):
    Dr = np.linalg.norm(D)
    u = D / Dr
    a = (u[0] ** 2 + u[1] ** 2) / (2 * sT * sT) + u[2] ** 2 / (2 * sL * sL)
    factor = q / Dr / (sT * sT * sL * np.sqrt(8 * np.pi ** 3))
    b = -((x - s[0]) / (sT * sT) * u[0] + (y - s[1]) / (sT * sT) * u[1] + (z - s[2]) / (sL * sL) * u[2])
    delta = ((x - s[0]) ** 2 + (y - s[1]) ** 2) / (2 * sT * sT) + (z - s[2]) ** 2 / (2 * sL * sL)
    sa2 = 2 * np.sqrt(a)
    integ = np.sqrt(np.pi) * (-erf(b / sa2) + erf((b + 2 * a * Dr) / sa2)) / sa2
    return factor * np.exp(b * b / (4 * a) - delta) * integ


def quad(x, y, z, s, D, sT, sL, npt, q=1.0):
    Dr = np.linalg.norm(D)
    u = D / Dr
    factor = q / Dr / (sT * sT * sL * np.sqrt(8 * np.pi ** 3))
    xs, ws = leggauss(npt)
    out = 0
    for sn, wn in zip(0.5 * Dr * (xs + 1), 0.5 * Dr * ws):
        out = out + wn * (np.exp(-(x - s[0] - sn * u[0]) ** 2 / (2 * sT * sT)) * np.exp(-(y - s[1] - sn * u[1]) ** 2 / (2 * sT * sT))
                          * np.exp(-(z - s[2] - sn * u[2]) ** 2 / (2 * sL * sL)))
    return factor * out


def trial(rng, ratio, npt):
    sT = rng.uniform(0.008, 0.06)
    sL = sT * np.sqrt(4 / 8.8)
    d = rng.normal(size=3)
    d /= np.linalg.norm(d)
    a = (d[0] ** 2 + d[1] ** 2) / (2 * sT * sT) + d[2] ** 2 / (2 * sL * sL)
    D = d * ratio / np.sqrt(2 * a)
    s = np.zeros(3)
    g = [np.linspace(min(0, D[k]) - 4 * w, max(0, D[k]) + 4 * w, 40) for k, w in enumerate((sT, sT, sL))]
    X, Y, Z = np.meshgrid(*g, indexing="ij")
    ref = closed(X, Y, Z, s, D, sT, sL)
    return np.abs(quad(X, Y, Z, s, D, sT, sL, npt) - ref).max() / ref.max()


if __name__ == "__main__":
    rng = np.random.default_rng(2)
    print("ratio  nodes needed for 1e-12   6 + 1.9 ratio")
    for ratio in (0.25, 0.5, 1, 2, 4, 8, 12, 16, 24, 32, 48, 64):
        for npt in range(2, 200):
            if max(trial(rng, ratio, npt) for _ in range(6)) < 1e-12:
                print(f"{ratio:5.2f}  {npt:4d}                     {6 + 1.9 * ratio:6.1f}")
                break
