"""
Synthetic Visual-Genome-shaped data: there is no dataset, no network and no pretrained weights in
the build / benchmark environment, so the drivers and the benchmark run on seeded random images with the VG entry
contract (dataloaders/visual_genome.py:187-197): 592x592 N(0,1) images, integer GT boxes in BOX_SCALE coordinates,
classes in 1..150, a handful of ground-truth relations per image.
"""
import numpy as np
import torch

from config import BOX_SCALE, IM_SCALE
from dataloaders.blob import Blob


class SyntheticVG(torch.utils.data.Dataset):
    def __init__(self, num_images=64, seed=1234, n_boxes=20, n_rels=30, num_classes=151, num_predicates=51,
                 im_size=IM_SCALE):
        self.num_images, self.seed, self.im_size = num_images, seed, im_size
        self.ind_to_classes = ['__background__'] + ['class%03d' % i for i in range(1, num_classes)]
        self.ind_to_predicates = ['__background__'] + ['pred%02d' % i for i in range(1, num_predicates)]
        rs = np.random.RandomState(seed)
        scale = BOX_SCALE / float(im_size)
        self.gt_boxes, self.gt_classes, self.relationships = [], [], []
        box_counts = n_boxes if isinstance(n_boxes, (list, tuple)) else [n_boxes] * num_images
        for im in range(num_images):
            n_boxes = int(box_counts[im % len(box_counts)])        # ragged object counts when a list is given
            boxes = np.zeros((0, 4))
            while boxes.shape[0] < n_boxes:                       # integer boxes in image space, de-duplicated
                x1y1 = rs.uniform(0, im_size - 32, (n_boxes, 2))
                wh = rs.uniform(16, 300, (n_boxes, 2))
                cand = np.round(np.concatenate((x1y1, np.minimum(x1y1 + wh, im_size - 1)), 1))
                boxes = np.unique(np.concatenate((boxes, cand), 0), axis=0)
                rs.shuffle(boxes)
            boxes = boxes[:n_boxes]
            self.gt_boxes.append((boxes * scale).astype(np.float32))          # stored at BOX_SCALE like VG
            self.gt_classes.append(rs.randint(1, num_classes, n_boxes).astype(np.int64))
            pairs = np.array([(i, j) for i in range(n_boxes) for j in range(n_boxes) if i != j])
            sel = pairs[rs.choice(len(pairs), size=min(n_rels, len(pairs)), replace=False)]
            self.relationships.append(np.column_stack((sel, rs.randint(1, num_predicates, sel.shape[0]))).astype(np.int64))

    @property
    def coco(self):
        from lib.evaluation.det_map import FauxCoco
        return FauxCoco(self.gt_classes, self.gt_boxes, len(self.ind_to_classes))

    @property
    def num_classes(self):
        return len(self.ind_to_classes)

    @property
    def num_predicates(self):
        return len(self.ind_to_predicates)

    def __len__(self):
        return self.num_images

    def __getitem__(self, index):
        g = torch.Generator().manual_seed(self.seed * 100003 + index)
        return {
            'img': torch.randn(3, self.im_size, self.im_size, generator=g),
            'img_size': (self.im_size, self.im_size, self.im_size / BOX_SCALE),
            'gt_boxes': self.gt_boxes[index].copy(),
            'gt_classes': self.gt_classes[index].copy(),
            'gt_relations': self.relationships[index].copy(),
            'scale': self.im_size / BOX_SCALE,
            'index': index,
            'flipped': False,
            'fn': 'synthetic_%06d' % index,
        }


def collate(entries, is_train, mode='rel'):
    blob = Blob(mode=mode, is_train=is_train, num_gpus=1, batch_size_per_gpu=len(entries))
    for d in entries:
        blob.append(d)
    blob.reduce()
    return blob


def make_blob(dataset, indices, is_train, mode='rel'):
    return collate([dataset[i] for i in indices], is_train, mode)


class SyntheticLoader(object):
    """rank-sharded iterator of Blobs (rank r takes batches r, r+world, ...)"""

    def __init__(self, dataset, batch_size, is_train, rank=0, world_size=1, mode='rel'):
        self.dataset, self.batch_size, self.is_train = dataset, batch_size, is_train
        self.rank, self.world_size, self.mode = rank, world_size, mode

    def __len__(self):
        return len(self.dataset) // (self.batch_size * self.world_size)

    def __iter__(self):
        for it in range(len(self)):
            start = (it * self.world_size + self.rank) * self.batch_size
            yield make_blob(self.dataset, range(start, start + self.batch_size), self.is_train, self.mode)
