"""Loop utilities (mirror of utils/utils.py:116-162)."""
import logging
import os

import torch


def mkdirs(path):
    if not os.path.exists(path):
        os.makedirs(path)


def save_checkpoint(state, save_path, is_best=False, filename='checkpoint.pth.tar'):
    """utils/utils.py:116-123"""
    if not os.path.exists(save_path):
        os.makedirs(save_path)
    if filename is not None:
        torch.save(state, os.path.join(save_path, filename))


def adadelta_eps_decay(optimizer, eps_decay):
    """utils/utils.py:134-140: multiplies eps of param group 0 ONLY and returns it."""
    for p in optimizer.param_groups:
        p['eps'] *= eps_decay
        logging.info('adadelta eps decayed to ' + str(p['eps']))
        return p['eps']
    return 0


class ScheSampleRampup(object):
    """utils/utils.py:143-162"""

    def __init__(self, start_epoch, final_epoch, final_rate):
        self.epoch = 0
        self.start_epoch, self.final_epoch, self.final_rate = start_epoch, final_epoch, final_rate
        self.linear = float(final_rate) / (final_epoch - start_epoch)

    def reset(self):
        self.epoch = 0

    def update(self, epoch):
        if epoch < self.start_epoch:
            return 0.0
        if epoch < self.final_epoch:
            return self.linear * (epoch - self.start_epoch)
        return self.final_rate


class AverageMeter(object):
    def __init__(self):
        self.reset()

    def reset(self):
        self.val = self.avg = self.sum = 0.0
        self.count = 0

    def update(self, val, n=1):
        self.val = val
        self.sum += val * n
        self.count += n
        self.avg = self.sum / self.count
