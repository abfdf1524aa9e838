"""ddsp_amd: the MI355X-native Harmonic + FilteredNoise synthesis path of magenta/ddsp.

    import ddsp_amd as ddsp
    harmonic = ddsp.synths.Harmonic(n_samples=64000, sample_rate=16000)
    audio = harmonic(amplitudes, harmonic_distribution, f0_hz)     # torch tensor in HBM
"""
from ddsp_amd import core
from ddsp_amd import dags
from ddsp_amd import effects
from ddsp_amd import losses
from ddsp_amd import processors
from ddsp_amd import synths
from ddsp_amd import spectral_ops

__version__ = '0.1.0'
