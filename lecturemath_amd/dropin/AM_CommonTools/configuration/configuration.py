"""KEY = value configuration reader, same contract as the reference's AM_CommonTools/configuration/configuration.py
(from_file :97-121, get :9-20, get_str/get_bool/get_int/get_float :22-45): keys upper-cased, '#' starts a comment,
lines without exactly one '=' are skipped, get() literal-evals and falls back to the raw string."""
import ast


class Configuration:
    def __init__(self, config_data, key_order=None):
        self.data = config_data
        self.key_order = key_order

    def get(self, name, default=None):
        if name not in self.data:
            return default
        try:
            return ast.literal_eval(self.data[name])
        except Exception:
            return self.data[name]

    def get_str(self, name, default=""):
        return self.data.get(name, default)

    def get_bool(self, name, default=False):
        return int(self.data[name]) > 0 if name in self.data else default

    def get_int(self, name, default=0):
        return int(self.data[name]) if name in self.data else default

    def get_float(self, name, default=0.0):
        return float(self.data[name]) if name in self.data else default

    def contains(self, name):
        return name in self.data

    def set(self, name, value):
        self.data[name] = value

    @staticmethod
    def from_file(filename):
        data, order = {}, []
        with open(filename, "r") as f:
            for line in f:
                line = line.split("#", 1)[0]
                parts = line.split("=")
                if len(parts) != 2:
                    continue
                key = parts[0].strip().upper()
                data[key] = parts[1].strip()
                order.append(key)
        return Configuration(data, order)
