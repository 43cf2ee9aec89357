"""Device-side construction of large synthetic channel batches (bench / large-scale tests plumbing).

`U` distinct, periodic symbol streams are built on the CPU with digiham_amd.synth, pulse-shaped
once on the GPU (circular FFT convolution with the TX root-raised-cosine), and fanned out to `B`
channels with per-channel circular delay, gain, DC offset and white noise.  The input of the
engine is therefore resident in HBM before any timed region starts.
"""
import numpy as np

from . import synth, _taps


def periodic_streams(proto, n_units, U, seed=1000):
    """U dibit streams of identical length whose frame grid wraps around seamlessly."""
    out = []
    for u in range(U):
        if proto == "dmr":
            s = synth.dmr_stream(seed + u, n_units, two_slots=(u % 2 == 0), lead_in=0)
        elif proto == "nxdn":
            s = synth.nxdn_stream(seed + u, n_units, lead_in=0)
        elif proto == "dstar":                       # n_units 96-bit frames of transmissions and noise gaps
            s = synth.dstar_stream(seed + u, n_units // 40 + 2, lead_in=0)[0]
            while len(s) < n_units * 96:
                s = np.concatenate([s, synth.dstar_stream(seed + u + 7777, n_units // 40 + 2, lead_in=0)[0]])
            s = s[:n_units * 96]
        elif proto == "pocsag":                      # n_units 32-bit words of transmissions and noise gaps
            s = synth.pocsag_stream(seed + u, n_units // 40 + 2, lead_in=0)[0]
            while len(s) < n_units * 32:
                s = np.concatenate([s, synth.pocsag_stream(seed + u + 7777, n_units // 40 + 2, lead_in=0)[0]])
            s = s[:n_units * 32]
        else:
            s = synth.ysf_stream(seed + u, n_units, mode="vd2", lead_in=0)
        out.append(s)
    n = min(len(s) for s in out)
    return np.stack([s[:n] for s in out])


def make_batch(torch, device, proto, B, n_units, U=64, seed=1000, sps=10, amplitude=0.5,
               snr_classes=(None, 20.0, 12.0), chunk=2048):
    """Returns (x [B][T] float32 on `device`, info dict)."""
    syms = periodic_streams(proto, n_units, min(U, B), seed)
    U = syms.shape[0]
    S = syms.shape[1]
    T = S * sps
    levels = (np.array([-1.0, 1.0], np.float32) if proto == "dstar" else                         # bit 1 above the centre
              np.array([1.0, -1.0], np.float32) if proto == "pocsag" else synth.LEVELS)          # POCSAG is received inverted
    lv = torch.tensor(levels, device=device)[torch.from_numpy(syms.astype(np.int64)).to(device)]      # [U][S]
    imp = torch.zeros((U, T), dtype=torch.float32, device=device)
    imp[:, ::sps] = lv
    if proto in ("dstar", "pocsag"):                     # NRZ with sloped edges (synth.fsk_shape), no TX RRC
        g = np.convolve(np.ones(sps), np.ones(5) / 5.0)
    else:
        g = (_taps.narrow() if proto == "nxdn" else _taps.wide()).astype(np.float64)
    g = (g / g.sum() * sps).astype(np.float32)
    gp = torch.zeros(T, dtype=torch.float32, device=device)
    gp[:len(g)] = torch.from_numpy(g).to(device)
    base = torch.fft.irfft(torch.fft.rfft(imp, dim=1) * torch.fft.rfft(gp)[None, :], n=T, dim=1)
    base = torch.roll(base, -(len(g) // 2), dims=1).to(torch.float32) * amplitude                             # [U][T]
    power = float((base.double() ** 2).mean().item())
    x = torch.empty((B, T), dtype=torch.float32, device=device)
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    burst = {"dmr": 144, "ysf": 480, "nxdn": 192, "dstar": 96, "pocsag": 32}[proto] * sps
    for c0 in range(0, B, chunk):
        c1 = min(B, c0 + chunk)
        for ch in range(c0, c1):
            shift = ((ch // U) * 7 * burst + (ch % sps)) % T          # whole bursts + a sub-symbol timing offset
            gain = (0.25, 0.5, 1.0, 2.0)[ch % 4]
            dc = (0.0, 0.05, -0.1, 0.2)[(ch // 4) % 4]
            x[ch] = torch.roll(base[ch % U], shift) * gain + dc
        sigma = torch.tensor([0.0 if snr_classes[ch % len(snr_classes)] is None else
                              (0.25, 0.5, 1.0, 2.0)[ch % 4] * np.sqrt(power / 10 ** (snr_classes[ch % len(snr_classes)] / 10))
                              for ch in range(c0, c1)], dtype=torch.float32, device=device)
        x[c0:c1] += sigma[:, None] * torch.randn((c1 - c0, T), dtype=torch.float32, device=device, generator=gen)
    return x, {"symbols_per_channel": S, "samples_per_channel": T, "unique_streams": U}
