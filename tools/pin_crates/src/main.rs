// Dumps what the crates behind colordetect and videocompare compute for the RGBA frames under tests/golden/ (64x48, raw) and for the
// seeded frames tests/frames.py generates (same splitmix64, restated in frames()): one JSON document on stdout.  Every call below is the
// call the reference makes (colordetect/imp.rs:68-79, videocompare/hashed_image.rs:24-79, :86-107).
use color_thief::{get_palette, ColorFormat};
use dssim_core::Dssim;
use image_hasher::{HashAlg, HasherConfig};
use rgb::FromSlice;
use std::{env, fs, path::Path};

fn splitmix64(state: &mut u64) -> u64 {
    *state = state.wrapping_add(0x9E37_79B9_7F4A_7C15);
    let mut z = *state;
    z = (z ^ (z >> 30)).wrapping_mul(0xBF58_476D_1CE4_E5B9);
    z = (z ^ (z >> 27)).wrapping_mul(0x94D0_49BB_1331_11EB);
    z ^ (z >> 31)
}

/// tests/frames.random_frame(seed, w, h): every byte iid uniform, eight bytes per splitmix64 word, little endian
fn random_frame(seed: u64, w: usize, h: usize) -> Vec<u8> {
    let mut s = seed;
    let mut out = Vec::with_capacity(w * h * 4 + 8);
    while out.len() < w * h * 4 {
        out.extend_from_slice(&splitmix64(&mut s).to_le_bytes());
    }
    out.truncate(w * h * 4);
    out
}

fn json_list<T: std::fmt::Display>(v: impl Iterator<Item = T>) -> String {
    format!("[{}]", v.map(|x| x.to_string()).collect::<Vec<_>>().join(", "))
}

fn main() {
    let dir = env::args().nth(1).unwrap_or_else(|| "../../tests/golden".into());
    let (w, h) = (64u32, 48u32);
    let mut frames: Vec<(String, Vec<u8>)> = ["red", "smpte", "snow"]
        .iter()
        .map(|n| {
            let p = Path::new(&dir).join(format!("videotestsrc_{n}_64x48_RGBA.bin"));
            (format!("videotestsrc_{n}"), fs::read(&p).unwrap_or_else(|e| panic!("{}: {e}", p.display())))
        })
        .collect();
    frames.push(("random_5EED0001_64x48".into(), random_frame(0x5EED_0001, w as usize, h as usize)));
    frames.push(("random_5EED0002_64x48".into(), random_frame(0x5EED_0002, w as usize, h as usize)));

    let algos = [("mean", HashAlg::Mean), ("gradient", HashAlg::Gradient), ("vertgradient", HashAlg::VertGradient),
                 ("doublegradient", HashAlg::DoubleGradient), ("blockhash", HashAlg::Blockhash)];
    let mut out = vec![];
    for (name, px) in &frames {
        let mut fields = vec![];
        for (q, n) in [(10u8, 2u8), (1, 8), (5, 5), (10, 255)] {
            // colordetect/imp.rs:68-74 -- the whole plane, ColorFormat of the negotiated caps
            let pal = get_palette(px, ColorFormat::Rgba, q, n).expect("get_palette");
            fields.push(format!("\"palette_q{q}_n{n}\": {}", json_list(pal.iter().map(|c| ((c.r as u32) << 16) | ((c.g as u32) << 8) | c.b as u32))));
            // colordetect/imp.rs:77-79
            let d = pal[0];
            fields.push(format!("\"name_q{q}_n{n}\": \"{}\"", color_name::css::Color::similar([d.r, d.g, d.b]).to_lowercase()));
        }
        let img = image::RgbaImage::from_raw(w, h, px.clone()).unwrap();
        for (an, alg) in &algos {
            // hashed_image.rs:104 (HasherConfig::new().hash_alg(algo).to_hasher()) and :37-45
            let hash = HasherConfig::new().hash_alg(*alg).to_hasher().hash_image(&img);
            fields.push(format!("\"{an}\": \"{}\"", hash.as_bytes().iter().map(|b| format!("{b:02x}")).collect::<String>()));
        }
        out.push(format!("  \"{name}\": {{{}}}", fields.join(", ")));
    }
    // pairwise distances: hashed_image.rs:70 (Hamming) and :72-75 (dssim)
    let dssim = Dssim::new();
    let mut pairs = vec![];
    for (i, (na, a)) in frames.iter().enumerate() {
        for (nb, b) in frames.iter().skip(i) {
            let ia = dssim.create_image_rgba(a.as_rgba(), w as usize, h as usize).unwrap();
            let ib = dssim.create_image_rgba(b.as_rgba(), w as usize, h as usize).unwrap();
            let (val, _) = dssim.compare(&ia, ib);
            let v: f64 = val.into();
            let img_a = image::RgbaImage::from_raw(w, h, a.clone()).unwrap();
            let img_b = image::RgbaImage::from_raw(w, h, b.clone()).unwrap();
            let mut f = vec![format!("\"dssim\": {v:e}")];
            for (an, alg) in &algos {
                let hs = HasherConfig::new().hash_alg(*alg).to_hasher();
                f.push(format!("\"{an}\": {}", hs.hash_image(&img_a).dist(&hs.hash_image(&img_b))));
            }
            pairs.push(format!("  \"{na}|{nb}\": {{{}}}", f.join(", ")));
        }
    }
    println!("{{\n \"crates\": \"color-thief 0.2.2, color-name 1.2.0, image 0.25.10, image_hasher 3.1.1, dssim-core 3.4.0\",\n \"frames\": {{\n{}\n }},\n \"pairs\": {{\n{}\n }}\n}}",
             out.join(",\n"), pairs.join(",\n"));
}
