"""The C-ABI library loads on a GPU-less host, exports every symbol include/ctcasr.h declares,
and rejects bad arguments before touching the device (no compute calls here)."""

import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def lib():
    from ctc_asr_amd import build, hip
    build.build(verbose=False)
    return hip.load()


def _declared_functions(header='ctcasr.h'):
    text = open(os.path.join(ROOT, 'include', header)).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(ctcasr_[a-z0-9_]+)\s*\(', text)))


def test_every_declared_symbol_is_exported_and_bound(lib):
    from ctc_asr_amd import hip
    declared = _declared_functions()
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(lib, name), name
    assert sorted(hip.SIGNATURES) == declared
    # the device ABI holds device entry points only: host helpers live in their own library
    assert not any('crc' in name or 'host' in name for name in declared)


def test_host_helper_library_exports_its_header(lib):
    from ctc_asr_amd import hostlib
    declared = _declared_functions('ctcasr_host.h')
    host = hostlib.load()
    for name in declared:
        assert hasattr(host, name), name
    assert sorted(hostlib.SIGNATURES) == declared
    assert hostlib.crc32c(b'123456789') == 0xE3069283


def test_version_and_error_strings(lib):
    assert lib.ctcasr_abi_version() == 7
    assert lib.ctcasr_error_string(0) == b'ok'
    assert b'workspace' in lib.ctcasr_error_string(-3)
    assert lib.ctcasr_error_string(-5) == b'in-kernel wait timed out'


def test_argument_errors_are_reported_not_thrown(lib):
    assert lib.ctcasr_log_softmax_fwd(None, None, 4, 29, None) == -1
    assert lib.ctcasr_ctc_loss_fwd_bwd(None, None, None, None, 5, 2, 29, 28, 3, 1.0, None, None,
                                       None, None, 0, None) == -1
    assert lib.ctcasr_rnn_fwd(2, None, None, None, None, None, 0, 2, 64, None, None, None, 0,
                              None) == -1
    # unknown flag bits of the step-range entry points are an argument error
    assert lib.ctcasr_rnn_fwd_steps(2, None, None, None, None, None, 8, 2, 64, None, None, None,
                                    None, 0, 0, 8, 64, None) == -1
    # ... the upper 24 bits are the residency ticket, not variant flags
    assert lib.ctcasr_rnn_fwd_steps(2, None, None, None, None, None, 8, 2, 64, None, None, None,
                                    None, 0, 0, 8, 5 << 8, None) == -1   # (null pointers, not flags)
    assert lib.ctcasr_rnn_resident_gate(None, 0, 2, 8, 2, 1024, 1, 200, None) == -3
    assert lib.ctcasr_rnn_resident_gate(None, 0, 2, 8, 2, 1024, 0, 200, None) == -1
    # the only process-wide option is the profiling switch
    assert lib.ctcasr_set_option(b'rnn_fwd_half_chip', 1) == -1
    assert lib.ctcasr_set_option(b'rnn_kernel_events', 0) == 0
    assert lib.ctcasr_adam_step(None, None, None, None, 10, 1e-3, 0.9, 0.999, 1e-8, 1, 1.0,
                                None, None) == -1
    # flag 64 is not a variant bit (16 = CTCASR_RNN_F16 is)
    assert lib.ctcasr_rnn_fwd_steps(2, None, None, None, None, None, 8, 2, 64, None, None, None,
                                    None, 0, 0, 8, 16, None) == -1       # (null pointers)
    assert lib.ctcasr_rnn_fwd_f16_supported(2, 8, 2, 64, 16) == 0
    assert lib.ctcasr_step_guard(None, None, 4, None, None, None, None, None) == -1
    assert lib.ctcasr_absmax(None, 4, None, None) == -1
    assert lib.ctcasr_colscale_from_max(None, 4, None, None, None) == -1
    assert lib.ctcasr_rnn_bwd_f16_supported(2, 8, 2, 64, 16) == 0
    # workspace sizing is pure host arithmetic
    need = lib.ctcasr_ctc_loss_workspace_bytes(500, 16, 29, 150)
    assert need >= 500 * 16 * 301 * 8 + 500 * 16 * 29 * 4
    assert lib.ctcasr_rnn_reserve_bytes(2, 500, 16, 1024) == 500 * 16 * 2 * 5 * 1024 * 4
    assert lib.ctcasr_rnn_workspace_bytes(2, 500, 16, 1024) > 6 * 16 * 1024 * 4


def test_host_wrappers_refuse_cpu_tensors(lib):
    import torch
    from ctc_asr_amd import hip
    with pytest.raises(hip.CtcAsrError):
        hip.log_softmax_fwd(torch.zeros(4, 29))
    from ctc_asr_amd.model import CTCModel, ModelConfig
    with pytest.raises(Exception):
        CTCModel(ModelConfig(num_units_rnn=64, num_layers_rnn=1, num_units_dense=32), 'cpu')


def test_probe_builds_cannot_stand_in_for_the_product(lib, monkeypatch):
    """The product library reports build flags 0.  A timing-probe build of the recurrence
    kernels (tools/build_alt.sh ... -DPRNN_PROBE_HALF_LOADS=1: skips half of the operand loads,
    results wrong on purpose) comes out of the same sources; `hip.load` must refuse it unless
    CTCASR_ALLOW_PROBE_BUILD=1 says that a probe is what the caller wants."""
    import subprocess
    from ctc_asr_amd import hip
    assert lib.ctcasr_build_flags() == 0
    out = subprocess.run(['bash', os.path.join(ROOT, 'tools', 'build_alt.sh'), 'abitest',
                          '-DPRNN_PROBE_HALF_LOADS=1', '-DPRNN_CHAIN_LB=8',
                          '-DPRNN_CHAIN_REGW=28'],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    path = os.path.join(ROOT, out.stdout.strip().splitlines()[-1])
    product = hip._lib
    try:
        monkeypatch.delenv('CTCASR_ALLOW_PROBE_BUILD', raising=False)
        with pytest.raises(hip.CtcAsrError, match='probe'):
            hip.load(path)
        assert hip._lib is product           # the refused library did not replace the product
        monkeypatch.setenv('CTCASR_ALLOW_PROBE_BUILD', '1')
        probe = hip.load(path)
        assert probe.ctcasr_build_flags() == (hip.BUILD_PROBE_WRONG_RESULTS |
                                              hip.BUILD_NONDEFAULT_TUNING)
    finally:
        hip._lib = product
        for ext in ('.so', '.o', '.remarks'):
            try:
                os.remove(path[:-3] + ext)
            except OSError:
                pass


def test_missing_library_fails_loudly(tmp_path):
    from ctc_asr_amd import hip
    with pytest.raises(hip.CtcAsrError):
        hip.load(str(tmp_path / 'libctcasr.so'))
