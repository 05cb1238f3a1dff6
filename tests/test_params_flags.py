import pytest

from ctc_asr_amd.params import FLAGS


def test_parse_like_absl():
    FLAGS.reset()
    rest = FLAGS.parse(['--batch_size=8', '--rnn_cell', 'lstm', '--nocudnn', '--delete',
                        '--conv_filters=32', '--conv_filters', '32', '--learning_rate=1e-4',
                        'positional', '--', '--dev'])
    assert rest == ['positional']
    assert FLAGS.batch_size == 8 and FLAGS.rnn_cell == 'lstm'
    assert FLAGS.cudnn is False and FLAGS.delete is True and FLAGS.dev is True
    assert FLAGS.conv_filters == [32, 32] and FLAGS.learning_rate == pytest.approx(1e-4)
    FLAGS.reset()
    assert FLAGS.batch_size == 16 and FLAGS.conv_filters == [32, 32, 96] and FLAGS.cudnn is True


def test_errors():
    FLAGS.reset()
    with pytest.raises(ValueError):
        FLAGS.parse(['--no_such_flag=1'])
    with pytest.raises(ValueError):
        FLAGS.parse(['--batch_size'])
    with pytest.raises(AttributeError):
        FLAGS.nope
    FLAGS.update(beam_width=64)
    assert FLAGS.beam_width == 64
    FLAGS.reset()
