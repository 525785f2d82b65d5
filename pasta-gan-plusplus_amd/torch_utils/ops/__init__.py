# MI355X-native operator modules with the reference's torch_utils.ops API.
