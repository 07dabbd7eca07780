#!/usr/bin/env python3
"""Regenerates integration/rust/cattus_hip.patch from a checkout of the reference (build container only):
    python integration/rust/make_patch.py [/root/reference]
Each edit below is anchored on a line of the reference; the output is the unified diff a maintainer applies with
`patch -p1 < cattus_hip.patch` at the top of the Cattus tree, after copying hip.rs to engine/src/net/hip.rs."""
import difflib
import sys
from pathlib import Path

REF = Path(sys.argv[1] if len(sys.argv) > 1 else "/root/reference")
OUT = Path(__file__).resolve().parent / "cattus_hip.patch"


def after(text: str, anchor: str, new: str) -> str:
    assert text.count(anchor) == 1, (anchor, text.count(anchor))
    return text.replace(anchor, anchor + new)


def before(text: str, anchor: str, new: str) -> str:
    assert text.count(anchor) == 1, (anchor, text.count(anchor))
    return text.replace(anchor, new + anchor)


def swap(text: str, old: str, new: str) -> str:
    assert text.count(old) == 1, (old, text.count(old))
    return text.replace(old, new)


EDITS = {}

EDITS["engine/Cargo.toml"] = lambda t: after(t, 'stockfish = ["pleco"]\n', "hip = []\n")

EDITS["engine/build.rs"] = lambda t: after(t, "fn main() {\n", '''    if cfg!(feature = "hip") {
        // libcattus_hip.so: python -m cattus_amd.build in the MI355X repository, CATTUS_HIP_LIB_DIR=<repo>/cattus_amd
        println!("cargo::rerun-if-env-changed=CATTUS_HIP_LIB_DIR");
        let dir = std::env::var("CATTUS_HIP_LIB_DIR").expect("CATTUS_HIP_LIB_DIR is not set, can't locate libcattus_hip.so");
        println!("cargo::rustc-link-search=native={dir}");
        println!("cargo::rustc-link-lib=dylib=cattus_hip");
        println!("cargo::rustc-link-arg=-Wl,-rpath,{dir}");
    }

''')

EDITS["engine/src/game/mod.rs"] = lambda t: after(t, "    fn set(&mut self, idx: usize, val: bool);\n", '''
    /// u64 words of one plane where planes leave the engine as raw words: the .traindata records and the HIP
    /// evaluator (chess 1, tic-tac-toe 1, hex 2: low half, high half)
    const PLANE_WORDS: usize;
    fn push_plane_words(&self, out: &mut Vec<u64>);
''')

EDITS["engine/src/chess/core.rs"] = lambda t: after(t, "impl Bitboard for ChessBitboard {\n    type Game = ChessGame;\n", '''    const PLANE_WORDS: usize = 1;
    fn push_plane_words(&self, out: &mut Vec<u64>) {
        out.push(self.get_raw());
    }
''')

EDITS["engine/src/hex/core.rs"] = lambda t: after(
    t, "impl<const BOARD_SIZE: usize> Bitboard for HexBitboard<BOARD_SIZE> {\n    type Game = HexGame<BOARD_SIZE>;\n", '''    const PLANE_WORDS: usize = 2;
    fn push_plane_words(&self, out: &mut Vec<u64>) {
        // low then high half, as training/self-play/src/serialize/hex.rs writes them
        out.push((self.get_raw() & 0xffffffffffffffff) as u64);
        out.push((self.get_raw() >> 64) as u64);
    }
''')

EDITS["engine/src/ttt/core.rs"] = lambda t: after(t, "impl Bitboard for TttBitboard {\n    type Game = TttGame;\n", '''    const PLANE_WORDS: usize = 1;
    fn push_plane_words(&self, out: &mut Vec<u64>) {
        out.push(self.get_raw() as u64);
    }
''')


def model_rs(t: str) -> str:
    t = after(t, "    Executorch,\n}\nimpl Default for InferenceConfig {", "")  # anchor check only
    t = swap(t, "    Executorch,\n}\nimpl Default for InferenceConfig {", '''    Executorch,
    /// MI355X leaf evaluator (libcattus_hip.so): {"engine": "hip", "device": 0, "dtype": "f16x2"}
    #[cfg(feature = "hip")]
    Hip {
        device: Option<i32>,
        dtype: Option<crate::net::hip::HipDtype>,
    },
}
impl Default for InferenceConfig {''')
    t = swap(t, "        output_names: Vec<String>,\n    },\n}\n", '''        output_names: Vec<String>,
    },
    /// the evaluator lives in NNetwork (it takes bitboard planes, not tensors): nothing to hold here
    #[cfg(feature = "hip")]
    Hip,
}
''')
    t = before(t, "            #[cfg(not(all(\n                feature = \"torch-python\",", '''            #[cfg(feature = "hip")]
            InferenceConfig::Hip { .. } => ModelImpl::Hip,
''')
    t = before(t, "            #[cfg(not(any(\n                feature = \"torch-python\",", '''            #[cfg(feature = "hip")]
            ModelImpl::Hip => panic!("the hip engine evaluates bitboard planes in NNetwork::evaluate_impl, not tensors"),
''')
    return t


EDITS["engine/src/net/model.rs"] = model_rs


def net_mod_rs(t: str) -> str:
    t = swap(t, "pub mod model;\n", '#[cfg(feature = "hip")]\npub mod hip;\npub mod model;\n')
    t = swap(t, "    batcher: Batcher<Vec<Game::Bitboard>, (Vec<f32>, f32)>,\n", '''    batcher: Batcher<Vec<Game::Bitboard>, (Vec<f32>, f32)>,
    #[cfg(feature = "hip")]
    hip: Option<hip::HipModel>,
''')
    t = swap(t, "        Self {\n            model: Mutex::new(Model::new(model_path, inference_cfg)),\n", '''        #[cfg(feature = "hip")]
        let hip = match inference_cfg {
            InferenceConfig::Hip { device, dtype } => {
                // planes per position: one evaluation's worth of planes of the initial position
                let planes_num = Game::PLANES_NUM;
                Some(hip::HipModel::new(
                    model_path.as_ref(),
                    batch_size,
                    planes_num,
                    <Game::Bitboard as Bitboard>::PLANE_WORDS,
                    Game::MOVES_NUM,
                    device.unwrap_or(0),
                    dtype.unwrap_or(hip::HipDtype::F16x2),
                ))
            }
            _ => None,
        };
        Self {
            #[cfg(feature = "hip")]
            hip,
            model: Mutex::new(Model::new(model_path, inference_cfg)),
''')
    t = swap(t, "        let planes = to_planes(pos);\n\n        let (move_scores, val) = self", '''        let planes = to_planes(pos);

        #[cfg(feature = "hip")]
        if let Some(hip) = &self.hip {
            // batching over the calling threads, planes_to_tensor, the network and the non-finite scrub: on the GPU
            let mut words = Vec::with_capacity(planes.len() * <Game::Bitboard as Bitboard>::PLANE_WORDS);
            for p in &planes {
                p.push_plane_words(&mut words);
            }
            let run_begin = Instant::now();
            let (move_scores, val) = hip.evaluate_planes(&words);
            {
                let mut metrics = self.metrics.lock().unwrap();
                metrics.activation_count.increment(1);
                metrics.run_duration.set(run_begin.elapsed().as_secs_f64());
            }
            let moves = pos.legal_moves().collect_vec();
            return (calc_moves_probs::<Game>(moves, &move_scores), val);
        }

        let (move_scores, val) = self''')
    return t


EDITS["engine/src/net/mod.rs"] = net_mod_rs

# planes per position: a constant next to MOVES_NUM (chess 18, hex 3, tic-tac-toe 3: */net*.rs position_to_planes)
GAME_CONST = {
    "engine/src/game/mod.rs": ("    const MOVES_NUM: usize;\n", "    /// planes position_to_planes returns for one position\n    const PLANES_NUM: usize;\n"),
    "engine/src/chess/core.rs": ("    const MOVES_NUM: usize = 1880;\n", "    const PLANES_NUM: usize = 18;\n"),
    "engine/src/hex/core.rs": ("    const MOVES_NUM: usize = BOARD_SIZE * BOARD_SIZE;\n", "    const PLANES_NUM: usize = 3;\n"),
    "engine/src/ttt/core.rs": ("    const MOVES_NUM: usize = Self::BOARD_SIZE * Self::BOARD_SIZE;\n", "    const PLANES_NUM: usize = 3;\n"),
}


def config_py(t: str) -> str:
    t = swap(t, "InferenceConfig = ExecutorchConfig | TorchPyConfig | OnnxTractConfig | OnnxOrtConfig\n", '''@dataclass(config={"extra": "forbid"}, kw_only=True)
class HipConfig:
    engine: Literal["hip"] = "hip"
    device: int = 0
    dtype: Literal["f16x2", "bf16", "f16", "f32"] = "f16x2"


InferenceConfig = ExecutorchConfig | TorchPyConfig | OnnxTractConfig | OnnxOrtConfig | HipConfig
''')
    return t


EDITS["training/cattus_train/config.py"] = config_py


def self_play_py(t: str) -> str:
    t = swap(t, "from cattus_train.config import ExecutorchConfig, InferenceConfig,", "from cattus_train.config import ExecutorchConfig, HipConfig, InferenceConfig,")
    t = swap(t, "        case OnnxOrtConfig():\n            features = [\"onnx-ort\"]\n", '''        case OnnxOrtConfig():
            features = ["onnx-ort"]
        case HipConfig():
            features = ["hip"]  # needs CATTUS_HIP_LIB_DIR in the environment (engine/build.rs)
''')
    t = swap(t, "        case OnnxOrtConfig() | OnnxTractConfig():\n            return \"onnx\"\n", '''        case OnnxOrtConfig() | OnnxTractConfig():
            return "onnx"
        case HipConfig():
            return "cattus"
''')
    t = swap(t, "        case OnnxOrtConfig() | OnnxTractConfig():  # onnx\n", '''        case HipConfig():  # flat blob of the state_dict; BatchNorm folding happens inside the evaluator
            from cattus_amd.weights import blob_from_module

            model_path.write_bytes(blob_from_module(model, input_shape))

        case OnnxOrtConfig() | OnnxTractConfig():  # onnx
''')
    return t


EDITS["training/cattus_train/self_play.py"] = self_play_py


def main():
    chunks = []
    for rel in sorted(EDITS):
        old = (REF / rel).read_text()
        new = EDITS[rel](old)
        if rel in GAME_CONST:
            anchor, add = GAME_CONST[rel]
            new = after(new, anchor, add)
        diff = difflib.unified_diff(old.splitlines(True), new.splitlines(True), f"a/{rel}", f"b/{rel}", n=2)
        chunks.append("".join(diff))
    header = (
        'Adds the MI355X leaf evaluator (libcattus_hip.so) to Cattus as inference engine "hip".\n'
        "Apply at the top of the Cattus tree after copying integration/rust/hip.rs to engine/src/net/hip.rs:\n"
        "    patch -p1 < cattus_hip.patch\n"
        "    CATTUS_HIP_LIB_DIR=<mi355x repo>/cattus_amd cargo build --release --features hip --bin chess_self_player\n"
        "Generated by integration/rust/make_patch.py against the reference snapshot of 2025-10-03.\n\n"
    )
    OUT.write_text(header + "".join(chunks))
    print(f"wrote {OUT} ({sum(c.count(chr(10)) for c in chunks)} lines)")


if __name__ == "__main__":
    main()
