"""Flag system with the reference's attribute names (options/base_options.py:12-164) plus the
options joint_train.py reads but upstream never defined (S3)."""
import argparse
import os

_STR, _INT, _FLT = str, int, float

# (flag, type, default) -- dashes and underscores exactly as upstream so that ``opt.<attr>`` matches
_FLAGS = [
    ('--works_dir', _STR, '.'), ('--dataroot', _STR, None), ('--dict_dir', _STR, ''), ('--gpu_ids', _STR, '0'),
    ('--name', _STR, 'vad'), ('--checkpoints_dir', _STR, './checkpoints'), ('--resume', _STR, ''),
    ('--enhance_resume', _STR, ''), ('--asr_resume', _STR, ''), ('--joint_resume', _STR, ''), ('--num_workers', _INT, 4),
    ('--feat_type', _STR, 'kaldi_magspec'), ('--left_context_width', _INT, 0), ('--right_context_width', _INT, 0),
    ('--delta_order', _INT, 0), ('--normalize_type', _INT, 1), ('--num_utt_cmvn', _INT, 20000),
    ('--num_utt_per_loading', _INT, 200), ('--lowSNR', _FLT, 5), ('--upSNR', _FLT, 30),
    ('--etype', _STR, 'vggblstmp'), ('--elayers', _INT, 4), ('--eunits', _INT, 320), ('--eprojs', _INT, 320),
    ('--subsample', _STR, '1_1_1_1_1'), ('--subsample-type', _STR, 'skip'),
    ('--atype', _STR, 'location'), ('--adim', _INT, 320), ('--aact-fuc', _STR, 'softmax'), ('--awin', _INT, 5), ('--aheads', _INT, 4),
    ('--aconv-chans', _INT, 10), ('--aconv-filts', _INT, 100),
    ('--dtype', _STR, 'lstm'), ('--dlayers', _INT, 1), ('--dunits', _INT, 300), ('--mtlalpha', _FLT, 0.5), ('--lsm-type', _STR, ''),
    ('--lsm-weight', _FLT, 0.0), ('--fusion', _STR, ''),
    ('--enhance_type', _STR, 'blstm'), ('--enhance_layers', _INT, 3), ('--enhance_units', _INT, 128), ('--enhance_projs', _INT, 128),
    ('--enhance_nonlinear_type', _STR, 'sigmoid'), ('--enhance_loss_type', _STR, 'L2'), ('--enhance_opt_type', _STR, 'gan_fbank'),
    ('--enhance_dropout_rate', _FLT, 0.0), ('--enhance_input_nc', _INT, 1), ('--enhance_output_nc', _INT, 1), ('--enhance_ngf', _INT, 64),
    ('--enhance_norm', _STR, 'batch'), ('--L1_loss_lambda', _FLT, 1.0),
    ('--gan_loss_lambda', _FLT, 1.0), ('--netD_type', _STR, 'basic'), ('--input_nc', _INT, 1), ('--n_layers_D', _INT, 3), ('--ndf', _INT, 64),
    ('--norm_D', _STR, 'batch'),
    ('--fbank_dim', _INT, 40), ('--fbank-opti-type', _STR, 'frozen'),
    ('--dropout-rate', _FLT, 0.0), ('--sche-samp-rate', _FLT, 0.0), ('--sche-samp-final-rate', _FLT, 0.6),
    ('--sche-samp-start-epoch', _INT, 5), ('--sche-samp-final_epoch', _INT, 15),
    ('--model-unit', _STR, 'char'), ('--space-loss-weight', _FLT, 0.1), ('--lmtype', _STR, None), ('--rnnlm', _STR, None),
    ('--kenlm', _STR, None), ('--word-rnnlm', _STR, None), ('--word-dict', _STR, None), ('--lm-weight', _FLT, 0.1),
    ('--batch-size', _INT, 30), ('--maxlen-in', _INT, 800), ('--maxlen-out', _INT, 150), ('--verbose', _INT, 1),
    # ---- S3: read by joint_train.py:122,146,168,171 but absent upstream (defaults fixed by the build) ----
    ('--enhance_loss_lambda', _FLT, 1.0), ('--coral_loss_lambda', _FLT, 0.0), ('--sche_samp_start_iter', _INT, 10 ** 9),
    ('--sche_samp_final_iter', _INT, 2 * 10 ** 9), ('--sche_samp_final_rate', _FLT, 0.6),
]
_BOOL_FLAGS = ['--mix_noise', '--no_lsgan', '--isGAN']


class BaseOptions(object):
    def __init__(self):
        self.parser = argparse.ArgumentParser(formatter_class=argparse.ArgumentDefaultsHelpFormatter)
        self.initialized = False

    def initialize(self):
        for flag, typ, default in _FLAGS:
            self.parser.add_argument(flag, type=typ, default=default)
        for flag in _BOOL_FLAGS:
            self.parser.add_argument(flag, action='store_true')
        self.parser.add_argument('--enhace_resume', dest='enhance_resume', type=str, help='alias: upstream typo (enhance_base_train.py:56)')
        self.initialized = True

    def parse(self, argv=None, save=True):
        if not self.initialized:
            self.initialize()
        self.opt = self.parser.parse_args(argv)
        self.opt.gpu_ids = [int(s) for s in str(self.opt.gpu_ids).split(',') if int(s) >= 0]     # base_options.py:133-138
        self.opt.mtl_mode = 'ctc' if self.opt.mtlalpha == 1.0 else ('att' if self.opt.mtlalpha == 0.0 else 'mtl')
        self.opt.labeldist = None
        exp_path = os.path.join(self.opt.checkpoints_dir, self.opt.name)
        self.opt.exp_path = exp_path
        if save:
            os.makedirs(exp_path, exist_ok=True)
            with open(os.path.join(exp_path, 'opt.txt'), 'wt') as f:
                f.write('------------ Options -------------\n')
                for k, v in sorted(vars(self.opt).items()):
                    f.write('%s: %s\n' % (str(k), str(v)))
                f.write('-------------- End ----------------\n')
        return self.opt
