"""Mirror of the reference's data_processing/volume_reader.py:36-53 (read_df / down_sample) on the native reader:
the payload arrives with one fread instead of one struct.unpack per float."""
import numpy as np

from .sample_io import df_read_payload


def down_sample(df, factor=2):
    """skimage.measure.block_reduce(df, (factor,)*3, np.mean): blocks that stick out are zero padded."""
    pad = [(0, (-s) % factor) for s in df.shape]
    p = np.pad(df, pad, mode="constant", constant_values=0)
    s = p.shape
    blocks = p.reshape(s[0] // factor, factor, s[1] // factor, factor, s[2] // factor, factor).transpose(0, 2, 4, 1, 3, 5)
    return np.mean(blocks, axis=(3, 4, 5))          # view_as_blocks + func(axis = the block axes), like skimage


def read_df(filename, scale_factor=1):
    payload, (X, Y, Z) = df_read_payload(filename)
    df = payload.reshape([X, Y, Z], order="F")
    if scale_factor != 1:
        df = down_sample(df, scale_factor)
    return df
