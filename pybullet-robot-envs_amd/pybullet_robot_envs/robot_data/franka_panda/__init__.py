import os


def get_data_path():
    return os.path.dirname(os.path.abspath(__file__))
