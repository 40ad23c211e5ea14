"""Held-out test videos for `train_vpd_model.py --no_test_video` (reference train_vpd_model.py:125-156,
action_dataset/eval.py:3-43): videos whose name starts with one of these prefixes are left out of student training,
so that downstream evaluation never sees a student that was distilled on its test videos.

fs / tennis: the reference's lists (data: the 2018 figure-skating competitions; four tennis matches, with the
per-player '', 'front__', 'back__' name forms of the tennis crop directories).  fx / diving48: derived from the
user's label files, at the paths the reference reads them from (finegym/data/gym99_val_element.txt: one
`<video>_E_..._A_... <label>` line per action, the prefix is the part before `_A_`; diving48/data/Diving48_V2_test.json:
a list of {'vid_name': ...}); VPD_GYM99_VAL_FILE / VPD_DIVING48_TEST_FILE override the paths."""
import json
import os

_HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GYM99_VAL_FILE = os.path.join(_HERE, 'finegym', 'data', 'gym99_val_element.txt')
DIVING48_V2_TEST_FILE = os.path.join(_HERE, 'diving48', 'data', 'Diving48_V2_test.json')

FS_TEST_PREFIXES = tuple('{}_short_program_2018'.format(ev) for ev in
                         ('men_olympic', 'men_world', 'women_olympic', 'women_world'))
TENNIS_TEST_VIDEOS = ('usopen_2015_mens_final_federer_djokovic', 'usopen_2019_womens_osaka_gauff',
                      'wimbledon_2019_mens_semifinal_federer_nadal', 'wimbledon_2019_womens_final_halep_williams')


def _need(path, flag_hint):
    if not os.path.isfile(path):
        raise FileNotFoundError('--no_test_video: {} not found ({})'.format(path, flag_hint))
    return path


def get_test_prefixes(dataset):
    if dataset.startswith('fs'):
        return FS_TEST_PREFIXES
    if dataset.startswith('tennis'):
        return tuple(side + v for side in ('', 'front__', 'back__') for v in TENNIS_TEST_VIDEOS)
    if dataset.startswith('fx'):
        path = _need(os.environ.get('VPD_GYM99_VAL_FILE', GYM99_VAL_FILE), 'the FineGym gym99 val label file; VPD_GYM99_VAL_FILE')
        with open(path) as fp:
            return tuple(line.split(' ')[0].split('_A_')[0] for line in fp if line.strip())
    if dataset.startswith('diving48'):
        path = _need(os.environ.get('VPD_DIVING48_TEST_FILE', DIVING48_V2_TEST_FILE), 'Diving48_V2_test.json; VPD_DIVING48_TEST_FILE')
        with open(path) as fp:
            return tuple(dict.fromkeys(a['vid_name'] for a in json.load(fp)))
    raise NotImplementedError('Unknown dataset: ' + dataset)
