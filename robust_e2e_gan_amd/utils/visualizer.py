"""Logger + running meters with the reference's call surface (utils/visualizer.py:49-245).
Plotting (matplotlib) is out of scope; the keys and cadence of the loop are preserved."""
import logging
import os

from .utils import AverageMeter, mkdirs


class Visualizer(object):
    def __init__(self, opt):
        self.opt = opt
        self.meters = {}
        self.log_path = os.path.join(getattr(opt, 'exp_path', '.'), 'main.log')
        mkdirs(os.path.dirname(self.log_path) or '.')
        self.logger = logging.getLogger('re2e.%s' % getattr(opt, 'name', 'run'))
        if not self.logger.handlers:
            self.logger.setLevel(logging.INFO)
            fh = logging.FileHandler(self.log_path)
            sh = logging.StreamHandler()
            fmt = logging.Formatter('%(asctime)s %(message)s')
            fh.setFormatter(fmt)
            sh.setFormatter(fmt)
            self.logger.addHandler(fh)
            self.logger.addHandler(sh)

    def get_logger(self):
        return self.logger

    def add_plot_report(self, keys, file_name):
        return {'keys': keys, 'file': file_name, 'history': []}

    def set_current_errors(self, errors):
        for k, v in errors.items():
            self.meters.setdefault(k, AverageMeter()).update(float(v))

    def get_current_errors(self, key):
        return self.meters[key].avg if key in self.meters else 0.0

    def print_current_errors(self, epoch, iters):
        msg = '(epoch: %d, iters: %d) ' % (epoch, iters) + ' '.join('%s: %.4f' % (k, m.avg) for k, m in sorted(self.meters.items())
                                                                       if k.startswith('train/'))
        self.logger.info(msg)

    def print_epoch_errors(self, epoch, iters):
        msg = '(epoch: %d, iters: %d) ' % (epoch, iters) + ' '.join('%s: %.4f' % (k, m.avg) for k, m in sorted(self.meters.items()))
        self.logger.info(msg)

    def plot_epoch_errors(self, epoch, iters, file_name):
        return {'file': file_name, 'epoch': epoch, 'iters': iters, 'values': {k: m.avg for k, m in self.meters.items()}}

    def plot_attention(self, att_w, dec_len, enc_len, file_name):
        return None

    def reset(self):
        for m in self.meters.values():
            m.reset()
