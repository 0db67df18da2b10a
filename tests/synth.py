"""Synthetic AIS-like GMSK IQ (SURVEY.md section 8d "Synthetic inputs").

Host-side numpy; used by the tests, by bench.py (which replicates a set of
unique channels on the device) and by the smoke test.  No reference code is
involved: this is an ordinary continuous-phase GMSK modulator (BT 0.4,
4-symbol Gaussian) driven by HDLC/NRZI framed random payloads.
"""
import numpy as np

SEED0 = 20260927
FS_BAUD = 9600.0


def crc16_hdlc(bits):
    """HDLC FCS (CRC-16/X.25) over a bit sequence given LSB-first per byte order."""
    crc = 0xFFFF
    for b in bits:
        x = (crc ^ int(b)) & 1
        crc >>= 1
        if x:
            crc ^= 0x8408
    crc ^= 0xFFFF
    return [(crc >> i) & 1 for i in range(16)]


def bit_stuff(bits):
    out, run = [], 0
    for b in bits:
        out.append(int(b))
        if b:
            run += 1
            if run == 5:
                out.append(0)
                run = 0
        else:
            run = 0
    return out


def nrzi_levels(bits, start_level=1):
    """AIS NRZI: a 0 toggles the level, a 1 keeps it.  Returns +-1 levels."""
    lvl, out = start_level, []
    for b in bits:
        if not b:
            lvl = -lvl
        out.append(lvl)
    return out


FLAG = [0, 1, 1, 1, 1, 1, 1, 0]


def sync_bits(family):
    """NRZ levels (as 0/1) of the correlator template's symbol sequence.
    'S': the stock template, bytes [1,1,0,0]*7 unpacked MSB first = 224 symbols
         (python/ais_demod.py:36-38 feeds packed bytes).
    'P': the intended 28-symbol preamble [1,1,0,0]*7 (python/ais.grc:106)."""
    if family == "S":
        out = []
        for byte in [1, 1, 0, 0] * 7:
            out.extend([(byte >> (7 - k)) & 1 for k in range(8)])
        return out
    if family == "P":
        return [1, 1, 0, 0] * 7
    raise ValueError(family)


def gaussian_pulse(osf, bt=0.4, span=4):
    """Gaussian frequency pulse convolved with a one-symbol rectangle, unit area."""
    n = span * osf
    t = (np.arange(n) + 1 - n / 2.0) / osf
    s = 2 * np.pi * bt / np.sqrt(np.log(2.0))
    g = np.exp(-0.5 * (s * t) ** 2)
    g /= g.sum()
    p = np.convolve(g, np.ones(osf))
    return p / p.sum()


def gmsk_waveform(levels, osf, bt=0.4):
    """Continuous-phase GMSK at osf samples/symbol for +-1 NRZ levels."""
    lv = np.repeat(np.zeros(0), 0)
    up = np.zeros(len(levels) * osf)
    up[::osf] = levels
    f = np.convolve(up, gaussian_pulse(osf, bt))  # per-sample phase increments (unit area per symbol)
    ph = np.cumsum(f) * (np.pi / 2)
    del lv
    return np.exp(1j * ph)


def make_burst(rng, family, sps, payload_bits=168, osf_mult=16, ramp_syms=8):
    """One burst at sps samples/symbol with a random fractional timing offset.
    Returns (iq complex128, info dict)."""
    payload = rng.integers(0, 2, payload_bits).tolist()
    frame = payload + crc16_hdlc(payload)
    stuffed = bit_stuff(frame)
    data_bits = FLAG + stuffed + FLAG
    sync = sync_bits(family)
    sync_lv = [1 if b else -1 for b in sync]
    ramp_lv = [(-1) ** k for k in range(ramp_syms)]
    data_lv = nrzi_levels(data_bits, start_level=sync_lv[-1])
    tail_lv = [data_lv[-1]] * 4
    levels = np.array(ramp_lv + sync_lv + data_lv + tail_lv, dtype=np.float64)
    osf = sps * osf_mult
    w = gmsk_waveform(levels, osf)
    frac = int(rng.integers(0, osf_mult))
    iq = w[frac::osf_mult]
    nsym_total = len(levels)
    iq = iq[: nsym_total * sps]
    # amplitude ramp over the ramp symbols
    env = np.ones(iq.size)
    r = ramp_syms * sps // 2
    env[:r] = np.linspace(0, 1, r, endpoint=False)
    env[-r:] = np.linspace(1, 0, r, endpoint=False)
    info = dict(data_bits=data_bits, payload=payload, frac=frac / osf_mult, sync_start_sym=ramp_syms,
                nsyms=nsym_total)
    return iq * env, info


def slot_samples(family, sps):
    return (512 if family == "S" else 320) * sps


def make_channel(seed, T, family="P", sps=4, amp=1.0, p_occ=0.5, ebn0_db=20.0, cfo_max=500.0, noise=True):
    """One channel of T samples.  Returns (complex64[T], list of burst infos with 'start')."""
    rng = np.random.default_rng(seed)
    fs = FS_BAUD * sps
    x = np.zeros(T, dtype=np.complex128)
    slot = slot_samples(family, sps)
    infos = []
    for s0 in range(0, T - slot + 1, slot):
        if rng.random() >= p_occ:
            continue
        iq, info = make_burst(rng, family, sps)
        cfo = rng.uniform(-cfo_max, cfo_max)
        ph = rng.uniform(-np.pi, np.pi)
        start = s0 + int(rng.integers(0, max(1, slot - iq.size)))
        n = np.arange(iq.size)
        x[start:start + iq.size] += amp * iq * np.exp(1j * (2 * np.pi * cfo / fs * n + ph))
        info.update(start=start, cfo=cfo, phase=ph, amp=amp)
        infos.append(info)
    if noise:
        # Eb/N0 with Eb = amp^2 * sps (energy per symbol in sample units)
        n0 = amp * amp * sps / (10 ** (ebn0_db / 10.0))
        sigma = np.sqrt(n0 / 2.0)
        x += rng.normal(0, sigma, T) + 1j * rng.normal(0, sigma, T)
    return x.astype(np.complex64), infos


def make_wideband(seed, nframes, lanes, fs=25e6, nlanes=1024, decim=512, amp=1.0, bursts_per_lane=3, cfo_max=300.0,
                  noise_sigma=1.0, gen_osf=40, group_delay=30113, tail_frames=1400):
    """BASELINE config 5's input: one wideband stream of nframes * decim samples at `fs` carrying
    family-S bursts (9600 baud GMSK, the stock 224-symbol sync segment) on the channel centres
    m * fs / nlanes of the given lanes, plus white noise.  Every burst is generated at gen_osf
    samples per symbol and carried to the wideband rate by interpolating its phase and envelope
    (constant-envelope CPM: both are smooth).  Bursts are placed so that, behind a channel filter
    of the given group delay, they end tail_frames items before the end of their lane's nframes
    output items (corr_est delays the stream by its template length before the bits come out).
    cfo_max: one bound for all lanes or {lane: bound}.
    Returns (complex64[nframes * decim], {lane: [burst info with 'start' in lane items]})."""
    rng = np.random.default_rng(seed)
    n = nframes * decim
    x = np.zeros(n, dtype=np.complex128)
    sps_w = fs / FS_BAUD  # wideband samples per symbol
    infos = {}
    for m in lanes:
        infos[m] = []
        f0 = m * fs / nlanes
        if f0 >= fs / 2:
            f0 -= fs
        seg = (n - group_delay - tail_frames * decim) // bursts_per_lane
        for b in range(bursts_per_lane):
            iq, info = make_burst(rng, "S", gen_osf, osf_mult=1)
            dur = int(info["nsyms"] * sps_w)
            if dur + 64 * decim > seg:
                raise ValueError("make_wideband: %d bursts of ~%d samples do not fit %d frames" % (bursts_per_lane, dur, nframes))
            lo = b * seg + int(rng.integers(0, max(1, seg - dur - 64 * decim)))
            t = np.arange(dur)
            tg = t * (gen_osf / sps_w)  # position on the generator's grid
            ph = np.interp(tg, np.arange(iq.size), np.unwrap(np.angle(iq)))
            env = np.interp(tg, np.arange(iq.size), np.abs(iq))
            cm = cfo_max[m] if isinstance(cfo_max, dict) else cfo_max
            cfo = rng.uniform(-cm, cm)
            ph0 = rng.uniform(-np.pi, np.pi)
            x[lo:lo + dur] += amp * env * np.exp(1j * (ph + 2 * np.pi * ((f0 + cfo) / fs) * (lo + t) + ph0))
            info.update(start=(lo + group_delay) / decim, cfo=cfo, phase=ph0, amp=amp, lane=m)
            infos[m].append(info)
    if noise_sigma > 0:
        x += rng.normal(0, noise_sigma / np.sqrt(2), n) + 1j * rng.normal(0, noise_sigma / np.sqrt(2), n)
    return x.astype(np.complex64), infos


def resampled_template(template_hi, osf_hi, sps):
    """The stock template, generated at osf_hi samples per symbol (modulate_vector_bc(gmsk_mod(osf_hi),
    ...)), carried to a fractional `sps` by interpolating its phase: what a receiver whose channel
    rate is not a whole multiple of the baud rate correlates against (corr_est_cc takes the template
    as an argument, lib/corr_est_cc_impl.cc:48-63)."""
    t = np.asarray(template_hi, dtype=np.complex128)
    n = int(np.floor(t.size / osf_hi * sps))
    ph = np.interp(np.arange(n) * (osf_hi / sps), np.arange(t.size), np.unwrap(np.angle(t)))
    return np.exp(1j * ph).astype(np.complex64)


def make_batch(nchan, T, family="P", sps=4, seed0=SEED0, **kw):
    """[nchan][T] complex64, channel-major, plus per-channel burst infos."""
    out = np.zeros((nchan, T), dtype=np.complex64)
    infos = []
    for c in range(nchan):
        out[c], inf = make_channel(seed0 + c, T, family, sps, **kw)
        infos.append(inf)
    return out, infos


def find_bits(hay, needle):
    """All start positions of the bit pattern `needle` in `hay` (uint8 arrays)."""
    hay = np.asarray(hay, dtype=np.uint8)
    needle = np.asarray(needle, dtype=np.uint8)
    if hay.size < needle.size:
        return []
    win = np.lib.stride_tricks.sliding_window_view(hay, needle.size)
    return np.nonzero((win == needle).all(axis=1))[0].tolist()
