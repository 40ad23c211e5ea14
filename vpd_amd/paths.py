"""Default data roots, identical to reference video_dataset_paths.py:4-23."""
import os

ROOT_DIR = 'data/sports'


def _triple(name):
    root = os.path.join(ROOT_DIR, name)
    return root, os.path.join(root, 'videos'), os.path.join(root, 'crops')


TENNIS_ROOT_DIR, TENNIS_VIDEO_DIR, TENNIS_CROP_DIR = _triple('tennis')
FS_ROOT_DIR, FS_VIDEO_DIR, FS_CROP_DIR = _triple('fs')
FX_ROOT_DIR, FX_VIDEO_DIR, FX_CROP_DIR = _triple('fx')
DIVING48_ROOT_DIR, DIVING48_VIDEO_DIR, DIVING48_CROP_DIR = _triple('diving48')

ROOT = {'tennis': TENNIS_ROOT_DIR, 'fs': FS_ROOT_DIR, 'fx': FX_ROOT_DIR, 'diving48': DIVING48_ROOT_DIR}
CROPS = {'tennis': TENNIS_CROP_DIR, 'fs': FS_CROP_DIR, 'fx': FX_CROP_DIR, 'diving48': DIVING48_CROP_DIR}
