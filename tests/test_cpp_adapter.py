"""The C++ host side: include/pebblegpu_steps.hpp (reference call shapes over the C ABI) and the Qt-free
FileSdrFeeder.  CPU tier: the header compiles against the library and the feeder delivers WAV frames with the
reference's scaling.  GPU tier: BASELINE config 1 end to end -- WAV -> feeder -> callback -> chain -> audio callback
-> compared with the oracle."""
import os
import struct
import subprocess

import numpy as np
import pytest

from tests.signals import lcg_noise

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "examples", "filesdr_chain")


def write_wav_pcm16(path, iq, fs):
    """16-bit PCM stereo RIFF/WAVE with a Pebble-style LIST/INFO chunk in front of the data (wavfile.cpp:66-140)."""
    pcm = np.empty(2 * len(iq), dtype="<i2")
    pcm[0::2] = np.round(iq.real * 32767.0).astype(np.int16)
    pcm[1::2] = np.round(iq.imag * 32767.0).astype(np.int16)
    data = pcm.tobytes()
    fmt = struct.pack("<HHIIHH", 1, 2, int(fs), int(fs) * 4, 4, 16)
    info = b"INFO" + b"lofr" + struct.pack("<I", 10) + b"100000000\x00" + b"mode" + struct.pack("<I", 2) + b"0\x00"
    chunks = b"fmt " + struct.pack("<I", len(fmt)) + fmt + b"LIST" + struct.pack("<I", len(info)) + info + (b"\x00" if len(info) & 1 else b"")
    chunks += b"data" + struct.pack("<I", len(data)) + data
    with open(path, "wb") as f:
        f.write(b"RIFF" + struct.pack("<I", 4 + len(chunks)) + b"WAVE" + chunks)
    return pcm


@pytest.fixture(scope="module")
def exe():
    import __graft_entry__ as g
    g.build()
    src = os.path.join(ROOT, "examples", "filesdr_chain.cpp")
    lib = os.path.join(ROOT, "pebblesdr_amd")
    if not os.path.exists(EXE) or os.path.getmtime(EXE) < max(os.path.getmtime(src), os.path.getmtime(os.path.join(ROOT, "include", "pebblegpu_steps.hpp"))):
        subprocess.check_call(["g++", "-std=c++14", "-O2", "-Wall", "-I" + os.path.join(ROOT, "include"), src, "-L" + lib, "-lpebblegpu",
                               "-Wl,-rpath," + lib, "-o", EXE])
    return EXE


def make_signal(nfr, fs=2048000, n=2048):
    t = np.arange(nfr * n) / fs
    x = 10 ** (-10 / 20) * (1 + 0.5 * np.cos(2 * np.pi * 1000 * t)) * np.exp(2j * np.pi * 100e3 * t)
    return x + lcg_noise(nfr * n, 1, 10 ** (-70 / 20))


def test_feeder_delivers_wav_frames_with_reference_scaling(exe, tmp_path):
    x = make_signal(5)[: 5 * 2048 - 100]  # ragged tail: only whole frames are delivered
    wav, out = str(tmp_path / "iq.wav"), str(tmp_path / "frames.bin")
    pcm = write_wav_pcm16(wav, x, 2048000)
    r = subprocess.run([exe, wav, "feed", out], capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout.strip() == "4 frames"
    got = np.fromfile(out, dtype=np.complex128)
    want = pcm[0::2][: 4 * 2048] / 32767.0 + 1j * (pcm[1::2][: 4 * 2048] / 32767.0)
    assert np.array_equal(got, want)


def test_feeder_rejects_non_wav(exe, tmp_path):
    p = tmp_path / "junk.wav"
    p.write_bytes(b"not a wave file at all")
    r = subprocess.run([exe, str(p), "feed", str(tmp_path / "o.bin")], capture_output=True, text=True)
    assert r.returncode == 1


@pytest.mark.gpu
@pytest.mark.parametrize("fft,taps", [(0, 0), (8192, 4097)])
def test_config1_wav_to_audio_through_cpp_host(exe, tmp_path, oracle_mod, fft, taps):
    """FileSDR-style WAV @ 2.048 Msps, AM, mixer +100 kHz, band-pass (-5000, 5000): stock 2048/1025 FastFIR and the
    '4096-tap' 8192/4097 variant BASELINE config 1 names."""
    nfr = 4 * 32
    x = make_signal(nfr)
    wav, out = str(tmp_path / "iq.wav"), str(tmp_path / "audio.bin")
    pcm = write_wav_pcm16(wav, x, 2048000)
    args = [exe, wav, "am", out, "100000", "-5000", "5000"] + ([str(fft), str(taps)] if fft else [])
    r = subprocess.run(args, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    got = np.fromfile(out, dtype=np.complex128)
    xq = pcm[0::2] / 32767.0 + 1j * (pcm[1::2] / 32767.0)
    ref = oracle_mod.Receiver(2048000, 2048, 4096, fft or 2048, taps or 1025)
    ref.set_mode(oracle_mod.AM); ref.set_mixer(100000); ref.set_filter(-5000, 5000)
    want = np.concatenate([ref.process(xq[f * 2048:(f + 1) * 2048], want_spectrum=False)[0] for f in range(nfr)])
    assert got.shape == want.shape and len(want) >= 3 * 2048
    err = np.sqrt(np.mean(np.abs(got - want) ** 2)) / np.sqrt(np.mean(np.abs(want) ** 2))
    assert err <= 1e-5


@pytest.mark.gpu
def test_plain_c_host_and_the_async_contract():
    """examples/async_contract.c (gcc, no Python wrapper): results read through pebblegpu_memcpy_d2h right behind a queued
    process call equal the synchronised ones, and an input refill right behind a call does not overtake it."""
    import __graft_entry__ as g
    g.build()
    src, exe_c = os.path.join(ROOT, "examples", "async_contract.c"), os.path.join(ROOT, "examples", "async_contract")
    lib = os.path.join(ROOT, "pebblesdr_amd")
    subprocess.check_call(["gcc", "-O2", "-Wall", "-I" + os.path.join(ROOT, "include"), src, "-L" + lib, "-lpebblegpu", "-Wl,-rpath," + lib, "-lm", "-o", exe_c])
    r = subprocess.run([exe_c], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip() == "ok", r.stdout + r.stderr
