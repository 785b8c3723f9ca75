"""chambers.models.backbones.vision_transformer on MI355X (reference:
chambers/models/backbones/vision_transformer.py:103-292 generic builder, :403-652 zoo, :655 preprocess_input).

`VisionTransformer(...)` returns a `Model` whose layers carry the reference's names, weight shapes and weight
order (patch_embeddings/embedding, add_cls_token, pos_embedding, encoder[.layers[i], .norm_layer], feature,
predictions — cf. test_units/manual_test_vit_weights.py:79-155), so timm / Keras weight mappings carry over via
get_weights()/set_weights().  Execution goes through the whole-model HIP engine (chambers_amd/engine.py).

`pooling` ('cls' | 'avg' | 'max' | 'sum' | None), `feature_dim` (tanh head) and `include_top` follow :172-191,272-283.
`DistilledVisionTransformer` / `DeiTS16` / `DeiTB16` (:295-400, :583-652) add the distillation token and its head.
The pretrained-weight names, default sizes and the feature-layer rule follow :99-128; the download itself is not built
(`weights="imagenet21k+_224"` needs the network): such a name resolves to the release's `.h5` in a local cache (`_load_weights`),
and `weights` may also be None or a path to a Keras `.h5` weight file (read by chambers_amd.utils.hdf5_lite) or an `.npz`."""
import os

import numpy as np
import torch

from ... import engine as E
from ... import kernels as K
from ..._keras_like import Layer, Sequential
from ...augmentations.image_augmentations import ImageNetNormalization
from ... import initializers
from ...layers.core import Conv2D, Dense, Dropout, Reshape
from ...layers.embedding import ConcatEmbedding, LearnedEmbedding1D
from ...layers.transformer import Encoder

# (model name -> `weights` name -> file suffix) of the reference's release table (vision_transformer.py:15-96).  The md5 hashes of
# that table are not carried: this build never downloads - a pretrained file is looked up in a local cache directory only.
_WEIGHT_SUFFIXES = {
    "vits16": {"imagenet_224_deit": "imagenet_1000_224_deit"},
    "vitb16": {"imagenet21k": "imagenet_21k_224", "imagenet21k+_224": "imagenet_21k_1000_224", "imagenet21k+_384": "imagenet_21k_1000_384",
               "imagenet_224_deit": "imagenet_21k_1000_224_deit", "imagenet_384_deit": "imagenet_21k_1000_384_deit"},
    "vitb32": {"imagenet21k": "imagenet_21k_224", "imagenet21k+_384": "imagenet_21k_1000_384"},
    "vitl16": {"imagenet21k": "imagenet_21k_224", "imagenet21k+_224": "imagenet_21k_1000_224", "imagenet21k+_384": "imagenet_21k_1000_384"},
    "vitl32": {"imagenet21k": "imagenet_21k_224", "imagenet21k+_384": "imagenet_21k_1000_384"},
    "deits16": {"imagenet_224": "imagenet_1000_224"},
    "deitb16": {"imagenet_224": "imagenet_1000_224", "imagenet_384": "imagenet_1000_384"},
}


def _are_weights_pretrained(weights, model_name):
    """vision_transformer.py:99-100: a (zoo model, weights name) pair of the release table."""
    return (model_name in _WEIGHT_SUFFIXES) and (weights in _WEIGHT_SUFFIXES[model_name])


def _get_model_info(weights, model_name):
    """vision_transformer.py:103-114: (default input size, has a pre-logits feature layer) from the file suffix."""
    if _are_weights_pretrained(weights, model_name):
        suffix = _WEIGHT_SUFFIXES[model_name][weights].replace("_deit", "")
        return int(suffix.split("_")[-1]), ("21k" in suffix and "1000" not in suffix)
    return 224, False


def _check_pretrained_shape(input_shape, default_size, weights, model_name):
    """vision_transformer.py:120-128."""
    if input_shape is not None and _are_weights_pretrained(weights, model_name):
        default_shape = (default_size, default_size, input_shape[-1])
        if tuple(input_shape) != default_shape:
            raise ValueError("Weights '{}' require `input_shape` to be {}.".format(weights, default_shape))


def _weights_cache_dir():
    return os.environ.get("CHB_WEIGHTS_DIR") or os.path.join(os.path.expanduser("~"), ".keras", "models")


def _load_weights(model, weights, include_top):
    """vision_transformer.py:149-169 without the download: a pretrained name resolves to `<model>_<suffix>[_no_top].h5` in the local
    cache ($CHB_WEIGHTS_DIR, else ~/.keras/models - where keras `get_file` would have put the release file), read by the pure-Python
    HDF5 reader (chambers_amd.utils.hdf5_lite); a `.npz` of Keras-named arrays with the same stem is accepted too; any other string
    is a path to an `.h5` / `.npz` file."""
    if _are_weights_pretrained(weights, model.name):
        stem = model.name + "_" + _WEIGHT_SUFFIXES[model.name][weights] + ("" if include_top else "_no_top")
        cands = [os.path.join(_weights_cache_dir(), stem + ext) for ext in (".h5", ".npz")]
        path = next((c for c in cands if os.path.exists(c)), None)
        if path is None:
            raise RuntimeError("pretrained weights %r of %s: the reference downloads %s.h5 from its GitHub release (vision_transformer.py:"
                               "149-167); this build has no network path - put that file (or %s.npz of Keras-named arrays) into %s, "
                               "or pass weights=None / a path to an .h5 or .npz file / use Model.load_timm_state_dict"
                               % (weights, model.name, stem, stem, _weights_cache_dir()))
        model.load_weights(path)
    elif weights is not None:
        if not os.path.exists(str(weights)):
            raise ValueError("weights file not found: %s" % (weights,))
        model.load_weights(weights)


def _obtain_input_shape(input_tensor, input_shape, default_size, min_size):
    """Static-shape validation of vision_transformer.py:117-146 (keras obtain_input_shape + full-shape check)."""
    if input_tensor is not None:
        input_shape = tuple(input_tensor.shape[1:])
    if input_shape is None:
        input_shape = (default_size, default_size, 3)
    input_shape = tuple(input_shape)
    if len(input_shape) != 3:
        raise ValueError("`input_shape` must be a tuple of three integers.")
    if None in input_shape:
        raise ValueError("Input shape must be fully specified; got input shape {}.".format(input_shape))
    if input_shape[-1] != 3:
        raise ValueError("The input must have 3 channels; got `input_shape=" + str(input_shape) + "`")
    if input_shape[0] < min_size or input_shape[1] < min_size:
        raise ValueError("Input size must be at least " + str(min_size) + "x" + str(min_size) + "; got `input_shape=" + str(input_shape) + "`")
    return input_shape


class Model(Layer):
    """The tf.keras.Model facade the reference's builder returns: named layers, weights, call/predict, plus
    `train_step` (what `fit` + chambers.optimizers.AdamW would run)."""

    def __init__(self, cfg, layers, name=None):
        super().__init__(name=name)
        self.cfg = cfg
        self._layers = layers
        self.built = True
        self._engines = {}
        self._loaded_version = {}
        self._dirty = None          # key of the training engine whose weights are newer than the layer variables
        self._decay_key = {}        # engine key -> id of the optimizer whose decay flags it carries

    def _sublayers(self):
        return list(self._layers)

    @property
    def layers(self):
        self._sync()
        return self._layers

    def get_layer(self, name):
        self._sync()          # a reader of a sub-layer's variables sees the trained weights, like keras after fit
        for layer in self._layers:
            if layer.name == name:
                return layer
        raise ValueError("No such layer: " + name)

    @property
    def input_shape(self):
        return (None,) + self.cfg.image_size + (3,)

    # ---- trained weights live in the training engine's flat buffer until somebody looks at the layer variables
    def _sync(self):
        """After train_step the newest weights are in the training engine (fp32 master buffer in HBM).  Every reader of the
        layer variables (call / predict, keras_weights, get_weights, save_weights, compile, the engines of other batch sizes)
        goes through here first, so `train_step` followed by `model(x)` behaves like Keras fit followed by predict.  The
        training engine itself is marked current afterwards: it keeps its Adam moments and step count."""
        key = self._dirty
        if key is None:
            return
        self._dirty = None
        self._assign(self._engines[key].export_keras_weights())
        self._loaded_version[key] = self._version

    @property
    def weights(self):
        self._sync()
        return Layer.weights.fget(self)

    trainable_weights = weights

    def set_weights(self, weights):
        self._sync()
        super().set_weights(weights)

    # ---- Keras-named weight dictionary <-> layer variables
    def keras_weights(self):
        self._sync()
        kw = {}
        emb = self.get_layer("patch_embeddings").get_layer("embedding")
        kw["patch_embeddings/embedding/kernel"], kw["patch_embeddings/embedding/bias"] = emb.kernel.numpy(), emb.bias.numpy()
        kw["add_cls_token/embeddings"] = self.get_layer("add_cls_token").embedding.numpy()
        if self.cfg.distilled:
            kw["add_dist_token/embeddings"] = self.get_layer("add_dist_token").embedding.numpy()
        kw["pos_embedding/embeddings"] = self.get_layer("pos_embedding").embedding.numpy()
        enc = self.get_layer("encoder")
        for i, l in enumerate(enc.layers):
            p = "encoder/layer_%d/" % i
            m = l.multi_head_attention
            for nm in ("w_query", "b_query", "w_value", "b_value", "w_key", "b_key", "w_projection", "b_projection"):
                kw[p + "multi_head_attention/" + nm] = getattr(m, nm).numpy()
            kw[p + "norm1/gamma"], kw[p + "norm1/beta"] = l.norm1.gamma.numpy(), l.norm1.beta.numpy()
            kw[p + "dense1/kernel"], kw[p + "dense1/bias"] = l.dense1.kernel.numpy(), l.dense1.bias.numpy()
            kw[p + "dense2/kernel"], kw[p + "dense2/bias"] = l.dense2.kernel.numpy(), l.dense2.bias.numpy()
            kw[p + "norm2/gamma"], kw[p + "norm2/beta"] = l.norm2.gamma.numpy(), l.norm2.beta.numpy()
        kw["encoder/norm/gamma"], kw["encoder/norm/beta"] = enc.norm_layer.gamma.numpy(), enc.norm_layer.beta.numpy()
        for nm in ("feature", "predictions", "predictions_dist"):
            try:
                l = self.get_layer(nm)
                kw[nm + "/kernel"], kw[nm + "/bias"] = l.kernel.numpy(), l.bias.numpy()
            except ValueError:
                pass
        return kw

    def assign_keras_weights(self, kw):
        self._dirty = None          # an explicit assignment supersedes whatever a training engine holds
        self._assign(kw)

    def _assign(self, kw):
        emb = self.get_layer("patch_embeddings").get_layer("embedding")
        emb.kernel.assign(kw["patch_embeddings/embedding/kernel"]); emb.bias.assign(kw["patch_embeddings/embedding/bias"])
        self.get_layer("add_cls_token").embedding.assign(kw["add_cls_token/embeddings"])
        if self.cfg.distilled:
            self.get_layer("add_dist_token").embedding.assign(kw["add_dist_token/embeddings"])
        self.get_layer("pos_embedding").embedding.assign(kw["pos_embedding/embeddings"])
        enc = self.get_layer("encoder")
        for i, l in enumerate(enc.layers):
            p = "encoder/layer_%d/" % i
            m = l.multi_head_attention
            for nm in ("w_query", "b_query", "w_value", "b_value", "w_key", "b_key", "w_projection", "b_projection"):
                getattr(m, nm).assign(kw[p + "multi_head_attention/" + nm])
            l.norm1.gamma.assign(kw[p + "norm1/gamma"]); l.norm1.beta.assign(kw[p + "norm1/beta"])
            l.dense1.kernel.assign(kw[p + "dense1/kernel"]); l.dense1.bias.assign(kw[p + "dense1/bias"])
            l.dense2.kernel.assign(kw[p + "dense2/kernel"]); l.dense2.bias.assign(kw[p + "dense2/bias"])
            l.norm2.gamma.assign(kw[p + "norm2/gamma"]); l.norm2.beta.assign(kw[p + "norm2/beta"])
        enc.norm_layer.gamma.assign(kw["encoder/norm/gamma"]); enc.norm_layer.beta.assign(kw["encoder/norm/beta"])
        for nm in ("feature", "predictions", "predictions_dist"):
            if nm + "/kernel" in kw:
                l = self.get_layer(nm)
                l.kernel.assign(kw[nm + "/kernel"]); l.bias.assign(kw[nm + "/bias"])
        self._bump()

    def load_weights(self, path):
        """keras Model.load_weights for the two file kinds this build reads: `.npz` of Keras-named arrays (what save_weights
        writes here) and Keras' own `.h5` / `.hdf5` weight files (the reference's pretrained releases, vision_transformer.py:149-169),
        read by the pure-Python HDF5 subset reader chambers_amd.utils.hdf5_lite (this image has no h5py)."""
        path = str(path)
        if path.endswith((".h5", ".hdf5", ".keras.h5")):
            self._load_keras_h5(path)
            return
        with np.load(path) as z:
            self.assign_keras_weights({k: z[k] for k in z.files})

    def _load_keras_h5(self, path):
        """keras hdf5_format.load_weights_from_hdf5_group (topological loading): the file's layers that have weights are matched IN
        ORDER with the model's layers that have weights, and each layer's arrays in the order of its `weight_names` with the layer's
        `weights` - names are not compared, counts and shapes are."""
        from ...utils.hdf5_lite import load_keras_weights
        self._sync()
        values, layout = load_keras_weights(path)
        file_layers = [(lname, names) for lname, names in layout if names]
        mine = [l for l in self._layers if l.weights]
        if len(file_layers) != len(mine):
            raise ValueError("You are trying to load a weight file containing %d layers into a model with %d layers."
                             % (len(file_layers), len(mine)))
        for (lname, names), layer in zip(file_layers, mine):
            ws = layer.weights
            if len(names) != len(ws):
                raise ValueError('Layer #%s (named "%s" in the current model) was found to correspond to layer %s in the save file. '
                                 "However the new layer %s expects %d weights, but the saved weights have %d elements."
                                 % (self._layers.index(layer), layer.name, lname, layer.name, len(ws), len(names)))
            layer.set_weights([values[n] for n in names])
        self._dirty = None
        self._bump()

    def load_timm_state_dict(self, state_dict):
        """Import a timm ViT `state_dict()` (or any name -> array mapping with timm's keys) through the conversion rules of
        test_units/manual_test_vit_weights.py:27-155 (chambers_amd.utils.weights)."""
        from ...utils.weights import timm_state_dict_to_keras
        self.assign_keras_weights(timm_state_dict_to_keras(state_dict, self.cfg.n_heads, include_top=self.cfg.include_top))

    def save_weights(self, path):
        """keras Model.save_weights: a path ending in `.h5` / `.hdf5` gets Keras' own HDF5 weight-file layout (what the reference's
        checkpoint callbacks write, callbacks.py:31-38,99,103; written by chambers_amd.utils.hdf5_lite, readable by h5py / Keras and
        by load_weights here); any other path an `.npz` of Keras-named arrays."""
        path = str(path)
        if path.endswith((".h5", ".hdf5")):
            from ...utils.hdf5_lite import save_keras_weights
            self._sync()
            save_keras_weights(path, [(l.name, [(n, v.numpy()) for n, v in l.named_weights()]) for l in self._layers])
            return
        np.savez(path, **self.keras_weights())

    # ---- execution
    def engine(self, batch_size, training=False, **kw):
        key = (int(batch_size), bool(training))
        if self._dirty is not None and self._dirty != key:
            self._sync()
        if key not in self._engines:
            self._engines[key] = E.ViTEngine(self.cfg, batch_size, training=training, **kw)
            self._loaded_version[key] = -1
        eng = self._engines[key]
        if self._loaded_version[key] != self._version:
            eng.load_keras_weights(self.keras_weights())
            self._loaded_version[key] = self._version
        return eng

    def call(self, inputs, training=None, **kwargs):
        """inputs: float32 NHWC already preprocessed (the reference model's own input), or uint8 NHWC, in which case
        `preprocess_input` (ImageNetNormalization 'tf') is fused in front.  Returns float32 logits / features."""
        eng = self.engine(inputs.shape[0], training=False)
        if inputs.dtype != torch.uint8:
            K.patchify_f32(inputs.to(torch.float32), self.cfg.patch_size, out=eng.patches)
            out = eng.forward(None, training=bool(training), prepatched=True)
        else:
            out = eng.forward(inputs, training=bool(training))
        soft = getattr(self, "classifier_activation", None) == "softmax"
        if isinstance(out, tuple):                        # distilled variant: [x_cls, x_dist] (vision_transformer.py:392-393)
            return [K.softmax_rows(K.cast_f32(o)) if soft else o.clone() for o in out]
        if soft:
            return K.softmax_rows(K.cast_f32(out))        # B x classes; training consumes logits (fused softmax-CE)
        return out.clone()

    predict = call

    def compile(self, optimizer=None, **unused):
        """keras Model.compile for the part the hot path uses: a chambers_amd.optimizers.AdamW (hyper-parameters may be
        schedules, e.g. chambers_amd.schedules.LinearWarmup; decay_include / decay_exclude select the decayed variables).
        The loss is the fused sparse softmax cross-entropy from logits."""
        self._sync()
        self.optimizer = optimizer
        self._decay_key = {}        # training engines keep weights, Adam moments and step count; their decay flags are re-derived

    def train_step(self, images_u8, labels, **opt):
        """One optimisation step on an (augmented) uint8 NHWC batch; returns the per-sample loss.  Uses the compiled
        optimizer if there is one, else AdamW with the keyword hyper-parameters given here."""
        optimizer = getattr(self, "optimizer", None)
        key = (int(images_u8.shape[0]), True)
        if optimizer is None:
            eng = self.engine(key[0], training=True)
            if self._decay_key.get(key) is not None:
                eng.set_decay_fn(None)
                self._decay_key[key] = None
            loss = eng.train_step(images_u8, labels, **opt)
            self._dirty = key
            return loss
        if opt:
            raise ValueError("hyper-parameters come from the compiled optimizer; got %s" % sorted(opt))
        if not self.cfg.include_top:
            raise ValueError("train_step needs the classification top (include_top=True)")
        eng = self.engine(key[0], training=True)
        # keyed on the filter lists themselves, not on the optimizer object: mutating decay_include / decay_exclude between steps
        # takes effect (optimizers.py:169-181 reads them at every apply)
        flt = (None if optimizer.decay_include is None else tuple(optimizer.decay_include),
               None if optimizer.decay_exclude is None else tuple(optimizer.decay_exclude))
        if self._decay_key.get(key) != flt:
            eng.set_decay_fn(optimizer.decay_fn(self.cfg))
            self._decay_key[key] = flt
        eng.forward(images_u8, training=True)
        loss = eng.loss(labels)
        eng.backward()
        optimizer.apply(eng)
        self._dirty = key
        return loss

    def sync_from_engine(self, batch_size=None):
        """Copy trained weights back into the layer variables (Keras layout).  Kept for callers of the round-1 API: every
        reader of the variables now does this lazily (`_sync`)."""
        self._sync()


def VisionTransformer(patch_size, patch_dim, n_encoder_layers, n_heads, ff_dim, dropout_rate=0.1, input_tensor=None, input_shape=None,
                      include_top=True, weights="imagenet21k+_224", pooling="cls", feature_dim=None, classes=1000,
                      classifier_activation=None, model_name=None):
    weights_are_pretrained = _are_weights_pretrained(weights, model_name)
    default_size, has_feature = _get_model_info(weights, model_name)
    if weights_are_pretrained and feature_dim is not None:
        raise ValueError("'weights' and 'feature_dim' are mutually exclusive.")
    elif weights_are_pretrained and has_feature:
        feature_dim = patch_dim
        if include_top:
            print("Warning: weights '{}' has no top. 'include_top' will be set to False.".format(weights))
            include_top = False
    _check_pretrained_shape(input_shape, default_size, weights, model_name)
    if classifier_activation not in (None, "linear", "softmax"):
        raise ValueError("classifier_activation must be None / 'linear' / 'softmax' (the loss kernel consumes logits); got %r"
                         % (classifier_activation,))
    shape = _obtain_input_shape(input_tensor, input_shape, default_size=default_size, min_size=patch_size)
    cfg = E.ViTConfig(patch_size, patch_dim, n_encoder_layers, n_heads, ff_dim, dropout_rate, image_size=shape[:2], classes=classes,
                      include_top=include_top, feature_dim=feature_dim, pooling="none" if pooling is None else pooling)
    tn = initializers.TruncatedNormal(stddev=0.02)
    patch_embeddings = Sequential([Conv2D(filters=patch_dim, kernel_size=patch_size, strides=patch_size, padding="valid", name="embedding"),
                                   Reshape([-1, patch_dim])], name="patch_embeddings")
    add_cls = ConcatEmbedding(n_embeddings=1, embedding_dim=patch_dim, side="left", axis=1, initializer=tn, name="add_cls_token")
    pos = LearnedEmbedding1D(initializer=tn, name="pos_embedding")
    drop = Dropout(dropout_rate)
    encoder = Encoder(embed_dim=patch_dim, num_heads=n_heads, ff_dim=ff_dim, num_layers=n_encoder_layers, attention_dropout_rate=dropout_rate,
                      dense_dropout_rate=dropout_rate, pre_norm=True, norm_output=True, name="encoder")
    n_tok = cfg.n_tokens
    for l in patch_embeddings.layers[:1]:
        l.build((None,) + shape); l.built = True
    add_cls.build((None, cfg.n_patches, patch_dim)); add_cls.built = True
    pos.build((None, n_tok, patch_dim)); pos.built = True
    encoder.build((None, n_tok, patch_dim)); encoder.built = True
    layers = [patch_embeddings, add_cls, pos, drop, encoder]
    feat_in = patch_dim
    if feature_dim is not None:
        f = Dense(units=feature_dim, activation="tanh", name="feature")
        f.build((None, patch_dim)); f.built = True
        layers.append(f)
        feat_in = feature_dim
    if include_top:
        head = Dense(units=classes, activation=None if classifier_activation == "linear" else classifier_activation, name="predictions")
        head.build((None, feat_in)); head.built = True
        layers.append(head)
    model = Model(cfg, layers, name=model_name)
    model.classifier_activation = classifier_activation if include_top else None
    _load_weights(model, weights, include_top)
    return model


def DistilledVisionTransformer(patch_size, patch_dim, n_encoder_layers, n_heads, ff_dim, dropout_rate=0.1, return_dist_token=True,
                               input_tensor=None, input_shape=None, include_top=True, weights="imagenet_224", pooling=None, classes=1000,
                               classifier_activation=None, model_name=None):
    """vision_transformer.py:295-400: ViT with a distillation token after the class token (`add_dist_token`, then `add_cls_token`,
    so the sequence is [cls, dist, patches]) and a second head `predictions_dist` on the distillation token.  Outputs
    `[x_cls, x_dist]`, or their average when return_dist_token=False.  (The reference's default `pooling=None` would feed the
    whole sequence to the class head; its zoo entries pass "cls", and so must callers here.)"""
    default_size, _has_feature = _get_model_info(weights, model_name)
    _check_pretrained_shape(input_shape, default_size, weights, model_name)
    if classifier_activation not in (None, "linear", "softmax"):
        raise ValueError("classifier_activation must be None / 'linear' / 'softmax'; got %r" % (classifier_activation,))
    if pooling is None:
        raise ValueError("pooling=None hands the whole token sequence to the class head in the reference (vision_transformer.py:373);"
                         " pass pooling='cls' (what DeiTS16 / DeiTB16 do) or 'avg' / 'max' / 'sum'")
    shape = _obtain_input_shape(input_tensor, input_shape, default_size=default_size, min_size=patch_size)
    cfg = E.ViTConfig(patch_size, patch_dim, n_encoder_layers, n_heads, ff_dim, dropout_rate, image_size=shape[:2], classes=classes,
                      include_top=include_top, pooling=pooling, distilled=True, return_dist_token=return_dist_token)
    tn = initializers.TruncatedNormal(stddev=0.02)
    patch_embeddings = Sequential([Conv2D(filters=patch_dim, kernel_size=patch_size, strides=patch_size, padding="valid", name="embedding"),
                                   Reshape([-1, patch_dim])], name="patch_embeddings")
    add_dist = ConcatEmbedding(n_embeddings=1, embedding_dim=patch_dim, side="left", axis=1, initializer=tn, name="add_dist_token")
    add_cls = ConcatEmbedding(n_embeddings=1, embedding_dim=patch_dim, side="left", axis=1, initializer=tn, name="add_cls_token")
    pos = LearnedEmbedding1D(initializer=tn, name="pos_embedding")
    encoder = Encoder(embed_dim=patch_dim, num_heads=n_heads, ff_dim=ff_dim, num_layers=n_encoder_layers, attention_dropout_rate=dropout_rate,
                      dense_dropout_rate=dropout_rate, pre_norm=True, norm_output=True, name="encoder")
    for l in patch_embeddings.layers[:1]:
        l.build((None,) + shape); l.built = True
    add_dist.build((None, cfg.n_patches, patch_dim)); add_dist.built = True
    add_cls.build((None, cfg.n_patches + 1, patch_dim)); add_cls.built = True
    pos.build((None, cfg.n_tokens, patch_dim)); pos.built = True
    encoder.build((None, cfg.n_tokens, patch_dim)); encoder.built = True
    layers = [patch_embeddings, add_dist, add_cls, pos, Dropout(dropout_rate), encoder]
    if include_top:
        for nm in ("predictions", "predictions_dist"):
            head = Dense(units=classes, activation=None if classifier_activation == "linear" else classifier_activation, name=nm)
            head.build((None, patch_dim)); head.built = True
            layers.append(head)
    model = Model(cfg, layers, name=model_name)
    model.classifier_activation = classifier_activation if include_top else None
    _load_weights(model, weights, include_top)
    return model


def _deit(model_name, patch_size, patch_dim, n_layers, n_heads, ff_dim):
    def build(return_dist_token=True, input_tensor=None, input_shape=None, include_top=True, weights="imagenet_224", pooling="cls",
              classes=1000, classifier_activation=None):
        return DistilledVisionTransformer(patch_size=patch_size, patch_dim=patch_dim, n_encoder_layers=n_layers, n_heads=n_heads,
                                          ff_dim=ff_dim, dropout_rate=0.1, return_dist_token=return_dist_token, input_tensor=input_tensor,
                                          input_shape=input_shape, include_top=include_top, weights=weights, pooling=pooling,
                                          classes=classes, classifier_activation=classifier_activation, model_name=model_name)
    build.__name__ = model_name
    return build


def _zoo(model_name, patch_size, patch_dim, n_layers, n_heads, ff_dim, default_weights):
    def build(input_tensor=None, input_shape=None, include_top=True, weights=default_weights, pooling="cls", feature_dim=None, classes=1000,
              classifier_activation=None):
        return VisionTransformer(patch_size=patch_size, patch_dim=patch_dim, n_encoder_layers=n_layers, n_heads=n_heads, ff_dim=ff_dim,
                                 dropout_rate=0.1, feature_dim=feature_dim, input_tensor=input_tensor, input_shape=input_shape,
                                 include_top=include_top, weights=weights, pooling=pooling, classes=classes,
                                 classifier_activation=classifier_activation, model_name=model_name)
    build.__name__ = model_name
    return build


# reference zoo constants: vision_transformer.py:403-652
ViTS16 = _zoo("vits16", 16, 384, 12, 6, 1536, "imagenet_224_deit")
ViTB16 = _zoo("vitb16", 16, 768, 12, 12, 3072, "imagenet21k+_224")
ViTB32 = _zoo("vitb32", 32, 768, 12, 12, 3072, "imagenet21k+_384")
ViTL16 = _zoo("vitl16", 16, 1024, 24, 16, 4096, "imagenet21k+_224")
ViTL32 = _zoo("vitl32", 32, 1024, 24, 16, 4096, "imagenet21k+_384")

DeiTS16 = _deit("deits16", 16, 384, 12, 6, 1536)      # vision_transformer.py:583-617
DeiTB16 = _deit("deitb16", 16, 768, 12, 12, 3072)     # :619-652

preprocess_input = ImageNetNormalization(mode="tf", name="vit_preprocess")   # vision_transformer.py:655
