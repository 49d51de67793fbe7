"""
Visual Genome dataset + loader with the reference's entry points (dataloaders/visual_genome.py:23-424): `VG`,
`VG.splits`, `load_graphs`, `load_image_filenames`, `load_info`, `vg_collate`, `VGDataLoader.splits`.

On-disk formats: `VG-SGG.h5` (split mask, boxes_1024 in (xc,yc,w,h), labels, img_to_first/last
box / rel, relationships, predicates), `VG-SGG-dicts.json`, `image_data.json`, the JPEGs.  The HDF5 container is read
with h5py when it is importable and with dataloaders/h5lite.py (pure Python, round 6) otherwise; the same arrays saved with
`numpy.savez` (`*.npz`) are accepted everywhere a `.h5` is.  The logic is pinned against the reference's own `load_graphs` /
`load_info` / `VG.__getitem__` (tests/test_vg_loader.py, goldens from tests/golden/make_golden.py), the container format
against files written by the real HDF5 library (tests/test_h5lite.py, tests/golden/make_vg_h5.py).

When the VG files are absent (this build / benchmark environment) `VG.splits` serves the synthetic VG-shaped dataset
(dataloaders/synthetic.py) with the same entry dict and attributes; the drivers do not change.

Data parallelism is one process per GPU: the loader shards the (per-epoch shuffled) index list by rank.
"""
import json
import os
from collections import defaultdict

import numpy as np
import torch
from torch.utils.data import Dataset

from config import VG_IMAGES, IM_DATA_FN, VG_SGG_FN, VG_SGG_DICT_FN, BOX_SCALE, IM_SCALE, PROPOSAL_FN
from dataloaders.blob import Blob
from dataloaders.synthetic import SyntheticVG, SyntheticLoader
from lib.fpn.box_intersections_cpu.bbox import bbox_overlaps


def _open_arrays(path):
    """mapping name -> array-like supporting [:] and fancy indexing: an h5py.File when h5py is importable, else the pure-Python
    reader of dataloaders/h5lite.py (the HDF5 subset VG-SGG.h5 and the proposal file use, pinned against files written by the
    real library: tests/test_h5lite.py); a numpy .npz with the same keys is accepted everywhere a .h5 is"""
    if str(path).endswith('.npz'):
        return np.load(path)
    try:
        import h5py
    except ImportError:
        from dataloaders import h5lite
        return h5lite.File(path, 'r')
    return h5py.File(path, 'r')


class VG(Dataset):
    def __init__(self, mode, roidb_file=VG_SGG_FN, dict_file=VG_SGG_DICT_FN, image_file=IM_DATA_FN, filter_empty_rels=True,
                 num_im=-1, num_val_im=5000, filter_duplicate_rels=True, filter_non_overlap=True, use_proposals=False,
                 image_dir=VG_IMAGES, expected_images=108073):
        if mode not in ('test', 'train', 'val'):
            raise ValueError("Mode must be in test, train, or val. Supplied {}".format(mode))
        self.mode = mode
        self.roidb_file, self.dict_file, self.image_file = roidb_file, dict_file, image_file
        self.filter_non_overlap = filter_non_overlap
        self.filter_duplicate_rels = filter_duplicate_rels and self.mode == 'train'
        self.split_mask, self.gt_boxes, self.gt_classes, self.relationships = load_graphs(
            self.roidb_file, self.mode, num_im, num_val_im=num_val_im, filter_empty_rels=filter_empty_rels,
            filter_non_overlap=self.filter_non_overlap and self.is_train)
        self.filenames = load_image_filenames(image_file, image_dir, expected=expected_images)
        self.filenames = [self.filenames[i] for i in np.where(self.split_mask)[0]]
        self.ind_to_classes, self.ind_to_predicates = load_info(dict_file)
        if use_proposals:
            p = _open_arrays(PROPOSAL_FN)
            rpn_rois, rpn_scores = p['rpn_rois'], p['rpn_scores']
            first = np.array(p['im_to_roi_idx'][:][self.split_mask])
            num = np.array(p['num_rois'][:][self.split_mask])
            self.rpn_rois = [np.column_stack((rpn_scores[first[i]:first[i] + num[i]], rpn_rois[first[i]:first[i] + num[i]]))
                             for i in range(len(self.filenames))]
        else:
            self.rpn_rois = None
        from dataloaders.image_transforms import SquarePad, Resize, ToTensor, Normalize, Compose
        self.transform_pipeline = Compose([SquarePad(), Resize(IM_SCALE), ToTensor(),
                                           Normalize(mean=[0.485, 0.456, 0.406], std=[0.229, 0.224, 0.225])])

    @property
    def is_train(self):
        return self.mode.startswith('train')

    @property
    def coco(self):
        """ground truth in the shape detection mAP is computed on (reference :103-127 builds a pycocotools COCO object;
        here lib/evaluation/det_map.py: same boxes [x, y, w+1, h+1], areas, annotation ids from 0)"""
        from lib.evaluation.det_map import FauxCoco
        return FauxCoco(self.gt_classes, self.gt_boxes, len(self.ind_to_classes))

    @classmethod
    def splits(cls, *args, **kwargs):
        """train / val / test datasets; the synthetic stand-in when the VG files are not on this machine"""
        if os.path.exists(kwargs.get('roidb_file', VG_SGG_FN)):
            kwargs.pop('seed', None)
            kwargs.pop('num_train_im', None)
            return cls('train', *args, **kwargs), cls('val', *args, **kwargs), cls('test', *args, **kwargs)
        seed = kwargs.get('seed', 1234)
        n_train = kwargs.get('num_train_im') if kwargs.get('num_train_im') is not None else 96
        n_val = max(1, min(kwargs.get('num_val_im', 5000), 24))
        return (SyntheticVG(num_images=n_train, seed=seed), SyntheticVG(num_images=n_val, seed=seed + 1),
                SyntheticVG(num_images=n_val, seed=seed + 2))

    def entry_geometry(self, index, image_size):
        """everything of __getitem__ that does not touch pixels: (flipped, gt_boxes, im_size, gt_rels) for an image
        of PIL size (w, h) -- the reference's arithmetic and numpy draw order (:147-194)"""
        w, h = image_size
        flipped = self.is_train and np.random.random() > 0.5
        gt_boxes = self.gt_boxes[index].copy()
        if self.is_train:      # crop boxes that are too large
            gt_boxes[:, [1, 3]] = gt_boxes[:, [1, 3]].clip(None, BOX_SCALE / max(w, h) * h)
            gt_boxes[:, [0, 2]] = gt_boxes[:, [0, 2]].clip(None, BOX_SCALE / max(w, h) * w)
        box_scale_factor = BOX_SCALE / max(w, h)
        if flipped:
            scaled_w = int(box_scale_factor * float(w))
            gt_boxes[:, [0, 2]] = scaled_w - gt_boxes[:, [2, 0]]
        img_scale_factor = IM_SCALE / max(w, h)
        if h > w:
            im_size = (IM_SCALE, int(w * img_scale_factor), img_scale_factor)
        elif h < w:
            im_size = (int(h * img_scale_factor), IM_SCALE, img_scale_factor)
        else:
            im_size = (IM_SCALE, IM_SCALE, img_scale_factor)
        gt_rels = self.relationships[index].copy()
        if self.filter_duplicate_rels:     # one predicate per (subject, object) pair, sampled
            assert self.mode == 'train'
            all_rel_sets = defaultdict(list)
            for (o0, o1, r) in gt_rels:
                all_rel_sets[(o0, o1)].append(r)
            gt_rels = np.array([(k[0], k[1], np.random.choice(v)) for k, v in all_rel_sets.items()])
        return flipped, gt_boxes, im_size, gt_rels

    def __getitem__(self, index):
        from PIL import Image
        image_unpadded = Image.open(self.filenames[index]).convert('RGB')
        flipped, gt_boxes, im_size, gt_rels = self.entry_geometry(index, image_unpadded.size)
        if flipped:
            image_unpadded = image_unpadded.transpose(Image.FLIP_LEFT_RIGHT)
        entry = {
            'img': self.transform_pipeline(image_unpadded),
            'img_size': im_size,
            'gt_boxes': gt_boxes,
            'gt_classes': self.gt_classes[index].copy(),
            'gt_relations': gt_rels,
            'scale': IM_SCALE / BOX_SCALE,
            'index': index,
            'flipped': flipped,
            'fn': self.filenames[index],
        }
        if self.rpn_rois is not None:
            entry['proposals'] = self.rpn_rois[index]
        assertion_checks(entry)
        return entry

    def __len__(self):
        return len(self.filenames)

    @property
    def num_predicates(self):
        return len(self.ind_to_predicates)

    @property
    def num_classes(self):
        return len(self.ind_to_classes)


def assertion_checks(entry):
    if len(tuple(entry['img'].size())) != 3:
        raise ValueError("Img must be dim-3")
    if entry['img'].size(0) != 3:
        raise ValueError("Must have 3 color channels")
    if entry['gt_classes'].shape[0] != entry['gt_boxes'].shape[0]:
        raise ValueError("GT classes and GT boxes must have same number of examples")
    assert (entry['gt_boxes'][:, 2] >= entry['gt_boxes'][:, 0]).all()
    assert (entry['gt_boxes'] >= -1).all()


def load_image_filenames(image_file, image_dir=VG_IMAGES, expected=108073):
    """image_data.json -> list of existing image paths, skipping the four corrupted files (reference :239-262).
    `expected` is the reference's sanity check on the full dataset (None to disable for subsets)."""
    with open(image_file, 'r') as f:
        im_data = json.load(f)
    corrupted_ims = ['1592.jpg', '1722.jpg', '4616.jpg', '4617.jpg']
    fns = []
    for img in im_data:
        basename = '{}.jpg'.format(img['image_id'])
        if basename in corrupted_ims:
            continue
        filename = os.path.join(image_dir, basename)
        if os.path.exists(filename):
            fns.append(filename)
    if expected is not None:
        assert len(fns) == expected, 'found %d images, expected %d' % (len(fns), expected)
    return fns


def _images_of_split(arrays, mode, num_im, num_val_im, need_rels):
    """indices of the images a split uses: split code (0 = train+val, 2 = test), must have boxes (and relations when
    `need_rels`), optional truncation, then the first `num_val_im` of the train pool form 'val'"""
    usable = (arrays['split'][:] == (2 if mode == 'test' else 0)) & (arrays['img_to_first_box'][:] >= 0)
    if need_rels:
        usable &= arrays['img_to_first_rel'][:] >= 0
    ids = np.flatnonzero(usable)
    if num_im > -1:
        ids = ids[:num_im]
    if num_val_im > 0 and mode == 'val':
        ids = ids[:num_val_im]
    elif num_val_im > 0 and mode == 'train':
        ids = ids[num_val_im:]
    return ids


def _corner_boxes(center_boxes):
    """(xc, yc, w, h) -> (x1, y1, x2, y2), in the container's integer arithmetic like the reference"""
    assert np.all(center_boxes[:, :2] >= 0) and np.all(center_boxes[:, 2:] > 0)
    out = center_boxes
    out[:, :2] = out[:, :2] - out[:, 2:] / 2
    out[:, 2:] = out[:, :2] + out[:, 2:]
    return out


def load_graphs(graphs_file, mode='train', num_im=-1, num_val_im=0, filter_empty_rels=True, filter_non_overlap=False):
    """GT boxes / classes / relations of one split from the VG-SGG container (behaviour of the reference's
    dataloaders/visual_genome.py:264-361, pinned by tests/test_vg_loader.py).
    :return: split_mask [num_images] bool, boxes (list of [n,4] x1,y1,x2,y2 at BOX_SCALE), gt_classes (list of [n]),
             relationships (list of [r,3]: box_ind_1, box_ind_2, predicate; box indices local to the image)"""
    if mode not in ('train', 'val', 'test'):
        raise ValueError('{} invalid'.format(mode))
    arrays = _open_arrays(graphs_file)
    ids = _images_of_split(arrays, mode, num_im, num_val_im, filter_empty_rels)
    split_mask = np.zeros(arrays['split'][:].shape[0], dtype=bool)
    split_mask[ids] = True

    labels = arrays['labels'][:, 0]
    corners = _corner_boxes(arrays['boxes_{}'.format(BOX_SCALE)][:])
    pairs = arrays['relationships'][:]
    predicates = arrays['predicates'][:, 0]
    assert pairs.shape[0] == predicates.shape[0]
    box_lo, box_hi = arrays['img_to_first_box'][:][ids], arrays['img_to_last_box'][:][ids]
    rel_lo, rel_hi = arrays['img_to_first_rel'][:][ids], arrays['img_to_last_rel'][:][ids]

    boxes, gt_classes, relationships = [], [], []
    for k, image in enumerate(ids):
        b0, b1 = box_lo[k], box_hi[k] + 1
        image_boxes = corners[b0:b1, :]
        if rel_lo[k] >= 0:
            local = pairs[rel_lo[k]:rel_hi[k] + 1] - b0                 # global box ids -> ids inside this image
            assert np.all(local >= 0) and np.all(local < image_boxes.shape[0])
            rels = np.column_stack((local, predicates[rel_lo[k]:rel_hi[k] + 1]))
        else:
            assert not filter_empty_rels
            rels = np.zeros((0, 3), dtype=np.int32)
        if filter_non_overlap:                                           # training on SGDet: keep touching pairs only
            assert mode == 'train'
            iou = bbox_overlaps(image_boxes, image_boxes)
            touching = np.flatnonzero(iou[rels[:, 0], rels[:, 1]] > 0.0)
            if touching.size == 0:
                split_mask[image] = False                                # an image without any such pair is dropped
                continue
            rels = rels[touching]
        boxes.append(image_boxes)
        gt_classes.append(labels[b0:b1])
        relationships.append(rels)
    return split_mask, boxes, gt_classes, relationships


def load_info(info_file):
    """VG-SGG-dicts.json -> (ind_to_classes, ind_to_predicates), background at index 0 (reference :364-380)"""
    info = json.load(open(info_file, 'r'))
    info['label_to_idx']['__background__'] = 0
    info['predicate_to_idx']['__background__'] = 0
    class_to_ind, predicate_to_ind = info['label_to_idx'], info['predicate_to_idx']
    return (sorted(class_to_ind, key=lambda k: class_to_ind[k]),
            sorted(predicate_to_ind, key=lambda k: predicate_to_ind[k]))


def vg_collate(data, num_gpus=1, is_train=False, mode='det'):
    assert mode in ('det', 'rel')
    blob = Blob(mode=mode, is_train=is_train, num_gpus=1, batch_size_per_gpu=len(data))
    for d in data:
        blob.append(d)
    blob.reduce()
    return blob


class _RankSampler(torch.utils.data.Sampler):
    """rank r of `world` takes every world-th batch of a (per-epoch shuffled) index list; all ranks draw the same
    permutation (seeded by epoch), so the shards are disjoint"""

    def __init__(self, n, batch_size, rank, world, shuffle, seed=0):
        self.n, self.bs, self.rank, self.world, self.shuffle, self.seed, self.epoch = n, batch_size, rank, world, shuffle, seed, 0

    def set_epoch(self, epoch):
        self.epoch = epoch

    def __len__(self):
        return (self.n // (self.bs * self.world)) * self.bs

    def __iter__(self):
        order = np.random.RandomState(self.seed + self.epoch).permutation(self.n) if self.shuffle else np.arange(self.n)
        nb = self.n // (self.bs * self.world)
        for b in range(nb):
            start = (b * self.world + self.rank) * self.bs
            for i in order[start:start + self.bs]:
                yield int(i)


class VGDataLoader(torch.utils.data.DataLoader):
    """Blobs of `batch_size` images per process (reference :395-424 without the per-GPU chunking: one process per GPU)"""

    @classmethod
    def splits(cls, train_data, val_data, batch_size=3, num_workers=1, num_gpus=1, mode='det', rank=0, world_size=1,
               **kwargs):
        assert mode in ('det', 'rel')
        if isinstance(train_data, SyntheticVG):        # synthetic stand-in: in-process iterator, no workers
            return (SyntheticLoader(train_data, batch_size, True, rank=rank, world_size=world_size, mode=mode),
                    SyntheticLoader(val_data, batch_size if mode == 'det' else 1, False, rank=rank,
                                    world_size=world_size, mode=mode))
        vb = batch_size if mode == 'det' else 1
        kwargs.setdefault('pin_memory', torch.cuda.is_available())      # Blob.pin_memory: asynchronous scatter()
        train_load = cls(dataset=train_data, batch_size=batch_size, num_workers=num_workers, drop_last=True,
                         sampler=_RankSampler(len(train_data), batch_size, rank, world_size, True),
                         collate_fn=lambda x: vg_collate(x, mode=mode, is_train=True), **kwargs)
        val_load = cls(dataset=val_data, batch_size=vb, num_workers=num_workers, drop_last=True,
                       sampler=_RankSampler(len(val_data), vb, rank, world_size, False),
                       collate_fn=lambda x: vg_collate(x, mode=mode, is_train=False), **kwargs)
        return train_load, val_load
