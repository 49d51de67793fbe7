"""
Configuration: the constants and command-line flags of the reference's config.py (config.py:33-61, :143-197),
so that `from config import ModelConfig, BOX_SCALE, IM_SCALE` keeps working in the drivers.
Dataset paths are taken from the environment (MOTIFS_DATA) instead of a hard-coded home directory.
"""
import os
from argparse import ArgumentParser

ROOT_PATH = os.path.dirname(os.path.realpath(__file__))
DATA_PATH = os.environ.get('MOTIFS_DATA', os.path.join(ROOT_PATH, 'data'))


def path(fn):
    return os.path.join(DATA_PATH, fn)


def stanford_path(fn):
    return os.path.join(DATA_PATH, 'stanford_filtered', fn)


VG_IMAGES = os.environ.get('MOTIFS_VG_IMAGES', path('VG_100K'))
RCNN_CHECKPOINT_FN = path('faster_rcnn_500k.h5')
IM_DATA_FN = stanford_path('image_data.json')
VG_SGG_FN = stanford_path('VG-SGG.h5')
VG_SGG_DICT_FN = stanford_path('VG-SGG-dicts.json')
PROPOSAL_FN = stanford_path('proposals.h5')
COCO_PATH = os.environ.get('MOTIFS_COCO', path('mscoco'))

MODES = ('sgdet', 'sgcls', 'predcls')

BOX_SCALE = 1024   # scale of the stored boxes
IM_SCALE = 592     # images are resized to this

BG_THRESH_HI = 0.5
BG_THRESH_LO = 0.0
RPN_POSITIVE_OVERLAP = 0.7
RPN_NEGATIVE_OVERLAP = 0.3
RPN_FG_FRACTION = 0.5
FG_FRACTION = 0.25
RPN_BATCHSIZE = 256
ROIS_PER_IMG = 256
REL_FG_FRACTION = 0.25
RELS_PER_IMG = 256
RELS_PER_IMG_REFINE = 64

BATCHNORM_MOMENTUM = 0.01
ANCHOR_SIZE = 16
ANCHOR_RATIOS = (0.23232838, 0.63365731, 1.28478321, 3.15089189)
ANCHOR_SCALES = (2.22152954, 4.12315647, 7.21692515, 12.60263013, 22.7102731)

_FLAGS = [
    # (flag, dest, kwargs)
    ('-coco', 'coco', dict(action='store_true', help='use COCO (deprecated in the reference)')),
    ('-ckpt', 'ckpt', dict(type=str, default='', help='checkpoint to load')),
    ('-det_ckpt', 'det_ckpt', dict(type=str, default='', help='detector checkpoint')),
    ('-save_dir', 'save_dir', dict(type=str, default='', help='where checkpoints go')),
    ('-ngpu', 'num_gpus', dict(type=int, default=3, help='number of GPUs (= ranks under torchrun)')),
    ('-nwork', 'num_workers', dict(type=int, default=1)),
    ('-lr', 'lr', dict(type=float, default=1e-3)),
    ('-b', 'batch_size', dict(type=int, default=2, help='images per GPU')),
    ('-val_size', 'val_size', dict(type=int, default=5000)),
    ('-l2', 'l2', dict(type=float, default=1e-4)),
    ('-clip', 'clip', dict(type=float, default=5.0)),
    ('-p', 'print_interval', dict(type=int, default=100)),
    ('-m', 'mode', dict(type=str, default='sgdet')),
    ('-model', 'model', dict(type=str, default='motifnet')),
    ('-old_feats', 'old_feats', dict(action='store_true')),
    ('-order', 'order', dict(type=str, default='confidence')),
    ('-cache', 'cache', dict(type=str, default='')),
    ('-gt_box', 'gt_box', dict(action='store_true')),
    ('-adam', 'adam', dict(action='store_true')),
    ('-test', 'test', dict(action='store_true')),
    ('-multipred', 'multi_pred', dict(action='store_true')),
    ('-nepoch', 'num_epochs', dict(type=int, default=25)),
    ('-resnet', 'use_resnet', dict(action='store_true')),
    ('-proposals', 'use_proposals', dict(action='store_true')),
    ('-nl_obj', 'nl_obj', dict(type=int, default=1)),
    ('-nl_edge', 'nl_edge', dict(type=int, default=2)),
    ('-hidden_dim', 'hidden_dim', dict(type=int, default=256)),
    ('-pooling_dim', 'pooling_dim', dict(type=int, default=4096)),
    ('-pass_in_obj_feats_to_decoder', 'pass_in_obj_feats_to_decoder', dict(action='store_true')),
    ('-pass_in_obj_feats_to_edge', 'pass_in_obj_feats_to_edge', dict(action='store_true')),
    ('-rec_dropout', 'rec_dropout', dict(type=float, default=0.1)),
    ('-use_bias', 'use_bias', dict(action='store_true')),
    ('-use_tanh', 'use_tanh', dict(action='store_true')),
    ('-limit_vision', 'limit_vision', dict(action='store_true')),
    # additions of this implementation (not in the reference)
    ('-synthetic', 'synthetic', dict(type=int, default=0, help='use N synthetic VG-shaped images (no dataset on disk)')),
    ('-seed', 'seed', dict(type=int, default=1234)),
    ('-max_iters', 'max_iters', dict(type=int, default=0, help='stop an epoch after this many batches (0 = all)')),
]


class ModelConfig(object):
    """argparse wrapper with the reference's single-dash flags; attributes are the `dest` names."""

    def __init__(self, argv=None):
        self.parser = self.setup_parser()
        self.args = vars(self.parser.parse_args(argv))
        print("~~~~~~~~ Hyperparameters used: ~~~~~~~")
        for k, v in self.args.items():
            print("{} : {}".format(k, v))
        self.__dict__.update(self.args)
        self.ckpt = os.path.join(ROOT_PATH, self.ckpt) if len(self.ckpt) != 0 else None
        self.cache = os.path.join(ROOT_PATH, self.cache) if len(self.cache) != 0 else None
        if len(self.save_dir) == 0:
            self.save_dir = None
        else:
            self.save_dir = os.path.join(ROOT_PATH, self.save_dir)
            os.makedirs(self.save_dir, exist_ok=True)
        assert self.val_size >= 0
        if self.mode not in MODES:
            raise ValueError("Invalid mode: mode must be in {}".format(MODES))
        if self.model not in ('motifnet', 'stanford'):
            raise ValueError("Invalid model {}".format(self.model))
        if self.ckpt is not None and not os.path.exists(self.ckpt):
            raise ValueError("Ckpt file ({}) doesnt exist".format(self.ckpt))

    @staticmethod
    def setup_parser():
        parser = ArgumentParser(description='training code')
        for flag, dest, kw in _FLAGS:
            parser.add_argument(flag, dest=dest, **kw)
        return parser
