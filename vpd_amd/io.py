"""File formats at the edge of the path (reference util/io.py:10-37, README.md:185-194):
json / pickle helpers with the reference's names and behaviour."""
import json
import pickle


def load_json(fpath):
    with open(fpath) as fp:
        return json.load(fp)


def store_json(fpath, obj):
    with open(fpath, 'w') as fp:
        json.dump(obj, fp)


def load_pickle(fpath):
    with open(fpath, 'rb') as fp:
        return pickle.load(fp)


def store_pickle(fpath, obj):
    with open(fpath, 'wb') as fp:
        pickle.dump(obj, fp)
