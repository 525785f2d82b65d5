# MI355X-native drop-in for the reference's torch_utils package (hot-path subset).
