"""Training flags (options/train_options.py:5-27)."""
from .base_options import BaseOptions


class TrainOptions(BaseOptions):
    def initialize(self):
        BaseOptions.initialize(self)
        for flag, typ, default in [('--opt_type', str, 'adadelta'), ('--lr', float, 0.005), ('--beta1', float, 0.5), ('--eps', float, 1e-8),
                                   ('--eps-decay', float, 0.01), ('--criterion', str, 'acc'), ('--threshold', float, 1e-4),
                                   ('--start_epoch', int, 0), ('--iters', int, 0), ('--epochs', int, 30), ('--shuffle_epoch', int, -1),
                                   ('--grad-clip', float, 5), ('--num-save-attention', int, 3), ('--num-saved-specgram', int, 3),
                                   ('--validate_freq', int, 8000), ('--print_freq', int, 500), ('--best_acc', float, 0),
                                   ('--best_loss', float, float('inf'))]:
            self.parser.add_argument(flag, type=typ, default=default)
