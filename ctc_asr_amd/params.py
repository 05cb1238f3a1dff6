"""Hyper-parameters, network layout and reporting options.

The flag *names and defaults* are the drop-in knob API of the reference
(``asr/params.py:15-134``; list in SURVEY.md section 8b).  The reference registers them with
``tf.flags``; here they live in a small self-contained registry (`FLAGS`) that parses the same
``--name=value`` / ``--name value`` / ``--[no]bool`` command lines, so scripts written against
``from asr.params import FLAGS`` keep working after changing the import.
"""

import os
import sys

import numpy as np

from ctc_asr_amd.labels import num_classes

BASE_PATH = os.path.realpath(os.path.join(os.path.dirname(os.path.realpath(__file__)), '../'))


class _Flag:
    __slots__ = ('name', 'kind', 'default', 'value', 'help')

    def __init__(self, name, kind, default, help_text):
        self.name, self.kind, self.default, self.help = name, kind, default, help_text
        self.value = list(default) if kind == 'multi_int' else default


class FlagValues:
    """Minimal absl-style flag container: attribute access, ``parse``, ``reset``."""

    def __init__(self):
        object.__setattr__(self, '_flags', {})

    # -- definition -------------------------------------------------------------------------
    def define(self, kind, name, default, help_text=''):
        if name in self._flags:
            raise ValueError('Duplicate flag "{}".'.format(name))
        self._flags[name] = _Flag(name, kind, default, help_text)

    # -- access -----------------------------------------------------------------------------
    def __getattr__(self, name):
        flags = object.__getattribute__(self, '_flags')
        if name not in flags:
            raise AttributeError('Unknown flag "{}".'.format(name))
        return flags[name].value

    def __setattr__(self, name, value):
        if name not in self._flags:
            raise AttributeError('Unknown flag "{}".'.format(name))
        self._flags[name].value = self._convert(self._flags[name], value)

    def __contains__(self, name):
        return name in self._flags

    def flag_values_dict(self):
        return {k: f.value for k, f in self._flags.items()}

    def defaults_dict(self):
        return {k: f.default for k, f in self._flags.items()}

    def reset(self):
        for flag in self._flags.values():
            flag.value = list(flag.default) if flag.kind == 'multi_int' else flag.default

    def update(self, **kwargs):
        for key, value in kwargs.items():
            setattr(self, key, value)
        return self

    # -- parsing ----------------------------------------------------------------------------
    @staticmethod
    def _convert(flag, raw):
        if flag.kind == 'string':
            return str(raw)
        if flag.kind == 'int':
            return int(raw)
        if flag.kind == 'float':
            return float(raw)
        if flag.kind == 'bool':
            if isinstance(raw, str):
                low = raw.lower()
                if low in ('1', 'true', 't', 'yes', 'y'):
                    return True
                if low in ('0', 'false', 'f', 'no', 'n'):
                    return False
                raise ValueError('Bad boolean "{}" for --{}.'.format(raw, flag.name))
            return bool(raw)
        if flag.kind == 'multi_int':
            if isinstance(raw, (list, tuple)):
                return [int(v) for v in raw]
            return [int(v) for v in str(raw).replace(',', ' ').split()]
        raise ValueError(flag.kind)

    def parse(self, argv=None):
        """Parse ``argv`` (without the program name); returns the unparsed remainder."""
        argv = list(sys.argv[1:] if argv is None else argv)
        rest, seen_multi = [], set()
        i = 0
        while i < len(argv):
            arg = argv[i]
            i += 1
            if arg == '--':
                continue
            if not arg.startswith('-'):
                rest.append(arg)
                continue
            body = arg.lstrip('-')
            name, eq, raw = body.partition('=')
            if name not in self._flags and name.startswith('no') and name[2:] in self._flags \
                    and self._flags[name[2:]].kind == 'bool':
                self._flags[name[2:]].value = False
                continue
            if name not in self._flags:
                raise ValueError('Unknown command line flag "{}".'.format(arg))
            flag = self._flags[name]
            if not eq:
                if flag.kind == 'bool':
                    flag.value = True
                    continue
                if i >= len(argv):
                    raise ValueError('Missing value for flag --{}.'.format(name))
                raw = argv[i]
                i += 1
            if flag.kind == 'multi_int':
                values = self._convert(flag, raw)
                if name in seen_multi:
                    flag.value = flag.value + values
                else:
                    flag.value = values
                    seen_multi.add(name)
            else:
                flag.value = self._convert(flag, raw)
        return rest


FLAGS = FlagValues()

# Directories (defaults are outside of the project directory, asr/params.py:13-27).
FLAGS.define('string', 'train_dir', os.path.join(BASE_PATH, '../ctc-asr-checkpoints/3c4r2d-rnn'),
             'Checkpoint / log directory (resume source unless --delete).')
_CORPUS_DIR = os.path.join(BASE_PATH, '../speech-corpus')
FLAGS.define('string', 'corpus_dir', os.path.join(_CORPUS_DIR, 'corpus'),
             'Root of the WAV corpus; CSV paths are relative to it.')
FLAGS.define('string', 'train_csv', os.path.join(_CORPUS_DIR, 'train.csv'), 'Training manifest (path;label;length).')
FLAGS.define('string', 'test_csv', os.path.join(_CORPUS_DIR, 'test.csv'), 'Test manifest.')
FLAGS.define('string', 'dev_csv', os.path.join(_CORPUS_DIR, 'dev.csv'), 'Validation manifest.')

# Layer and activation options (asr/params.py:29-50).
FLAGS.define('string', 'used_model', 'ds2', "Front-end: 'ds1' = 3 dense layers, 'ds2' = 2-D convolutions.")
FLAGS.define('int', 'num_units_dense', 2048, 'Width of every dense layer.')
FLAGS.define('float', 'relu_cutoff', 20.0, 'Upper clip applied after each ReLU.')
FLAGS.define('multi_int', 'conv_filters', [32, 32, 96],
             'Number of filters for each convolutional layer (3 = reference stack; 2 entries '
             'select the "2-conv" variant of BASELINE.json).')
FLAGS.define('int', 'num_layers_rnn', 4, 'Depth of the bidirectional recurrent stack.')
FLAGS.define('int', 'num_units_rnn', 2048, 'Hidden units per direction.')
FLAGS.define('string', 'rnn_cell', 'rnn_relu', "Recurrent cell: rnn_relu | rnn_tanh | lstm | gru.")

# Inputs (asr/params.py:52-61).
FLAGS.define('int', 'batch_size', 16, 'Utterances per minibatch (per GPU when data parallel).')
FLAGS.define('string', 'feature_type', 'mfcc', "'mel' = 80 log-mel bands, 'mfcc' = 40 cepstra + 40 deltas.")
FLAGS.define('string', 'feature_normalization', 'local', "Per-utterance normalisation: none | local | local_scalar.")
FLAGS.define('bool', 'features_drop_every_second_frame', False,
             'Keep only every second feature frame (Deep Speech 1 style).')

# Learning rate (asr/params.py:63-74; the three decay flags are inert in the reference too).
FLAGS.define('int', 'max_epochs', 15, 'Total epochs (the first one walks the CSV in order).')
FLAGS.define('float', 'learning_rate', 1e-5, 'Adam step size.')
FLAGS.define('float', 'learning_rate_decay_factor', 4 / 5, 'Accepted, inert (as in the reference).')
FLAGS.define('int', 'steps_per_decay', 75000, 'Accepted, inert.')
FLAGS.define('float', 'minimum_lr', 1e-6, 'Accepted, inert. ')

# Adam (asr/params.py:76-82).
FLAGS.define('float', 'adam_beta1', 0.9, 'First-moment decay.')
FLAGS.define('float', 'adam_beta2', 0.999, 'Second-moment decay.')
FLAGS.define('float', 'adam_epsilon', 1e-8, 'Added to sqrt(v) (TensorFlow form).')

# CTC decoder (asr/params.py:84-86).
FLAGS.define('int', 'beam_width', 1024, 'Leaves kept by the CTC beam search (<= 1024).')

# Dropout (asr/params.py:88-94).
FLAGS.define('float', 'conv_dropout_rate', 0.0, 'Drop probability after each conv layer.')
FLAGS.define('float', 'rnn_dropout_rate', 0.0, 'Drop probability between recurrent layers.')
FLAGS.define('float', 'dense_dropout_rate', 0.1, 'Drop probability after each dense layer.')

# Corpus (asr/params.py:96-103).
FLAGS.define('int', 'num_buckets', 96, 'Upper bound on length buckets.')
FLAGS.define('int', 'num_classes', num_classes(), 'Alphabet size + unused id 0 + CTC blank.')
FLAGS.define('int', 'sampling_rate', 16000, 'Expected WAV sampling rate in Hz.')

# Performance / GPU (asr/params.py:105-112).  `cudnn=True` selects the fused recurrent kernels
# with cuDNN semantics (no sequence lengths); False selects the length-aware tanh-RNN semantics of
# the TensorFlow BasicRNNCell path.  Both run on the MI355X HIP kernels.
FLAGS.define('bool', 'cudnn', True, 'cuDNN-semantics RNN stack (True) or TF BasicRNNCell (False).')
FLAGS.define('int', 'shuffle_buffer_size', 2 ** 14, 'Sliding shuffle window of the bucketed targets.')

# Logging (asr/params.py:114-125).
FLAGS.define('int', 'log_frequency', 200, 'Steps between loss / throughput log lines.')
FLAGS.define('int', 'num_samples_to_report', 4, 'Decoded examples printed per evaluation.')
FLAGS.define('int', 'gpu_hook_query_frequency', 5, 'Accepted for compatibility (NVML hook of the reference).')
FLAGS.define('int', 'gpu_hook_average_queries', 100, 'Accepted for compatibility. ')

# Miscellaneous (asr/params.py:127-137).
FLAGS.define('bool', 'delete', False, 'Wipe train_dir first instead of resuming.')
FLAGS.define('int', 'random_seed', 0, 'Seed for init / dropout / shuffling; 0 = wall clock.')
FLAGS.define('bool', 'log_device_placement', False, 'Accepted for compatibility (no effect).')
FLAGS.define('bool', 'allow_vram_growth', True, 'Accepted for compatibility (no effect).')

# Driver-specific flags of the reference: `dev` (asr/evaluate.py:10), `input` (asr/predict.py:13).
FLAGS.define('bool', 'dev', False, 'evaluate.py: score dev.csv instead of test.csv.')
FLAGS.define('string', 'input', '', 'predict.py: WAV file to decode.')

# ####### Constants (asr/params.py:138-155). #########
NP_FLOAT = np.float32

MIN_EXAMPLE_LENGTH = 0.7
MAX_EXAMPLE_LENGTH = 17.0

WIN_LENGTH = 0.025  # Window length in seconds.
WIN_STEP = 0.010  # Step between successive windows in seconds.
NUM_FEATURES = 80  # Number of features to extract.

CSV_HEADER_PATH = 'path'
CSV_HEADER_LABEL = 'label'
CSV_HEADER_LENGTH = 'length'
CSV_FIELDNAMES = [CSV_HEADER_PATH, CSV_HEADER_LABEL, CSV_HEADER_LENGTH]
CSV_DELIMITER = ';'


def get_parameters():
    """Summary string of training and network parameters (``asr/params.py:160-184``)."""
    rows = [
        '',
        '\tLearning Rate (lr={}, steps_per_decay={:,d}, decay_factor={});'.format(
            FLAGS.learning_rate, FLAGS.steps_per_decay, FLAGS.learning_rate_decay_factor),
        '\tGPU-Options (cudnn={});'.format(FLAGS.cudnn),
        '\tModel (used_model={}, beam_width={:,d})'.format(FLAGS.used_model, FLAGS.beam_width),
        '\tConv (conv_filters={}); Dense (num_units={:,d});'.format(
            FLAGS.conv_filters, FLAGS.num_units_dense),
        '\tRNN (num_units={:,d}, num_layers={:,d});'.format(
            FLAGS.num_units_rnn, FLAGS.num_layers_rnn),
        '\tTraining (batch_size={:,d}, max_epochs={:,d}, log_frequency={:,d});'.format(
            FLAGS.batch_size, FLAGS.max_epochs, FLAGS.log_frequency),
        '\tFeatures (type={}, normalization={}, skip_every_2nd_frame={});'.format(
            FLAGS.feature_type, FLAGS.feature_normalization,
            FLAGS.features_drop_every_second_frame),
    ]
    return '\n'.join(rows)
