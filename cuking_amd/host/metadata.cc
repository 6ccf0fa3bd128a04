#include "metadata.h"

#include <cmath>
#include <cstdlib>
#include <fstream>
#include <sstream>

namespace cuking_host {

namespace {

// Recursive-descent JSON reader that keeps only what the metadata needs:
// top-level "num_sites" and "samples"; everything else is validated and skipped.
class Parser {
 public:
  explicit Parser(const std::string &text) : s_(text) {}

  std::string Run(Metadata *out) {
    SkipWs();
    if (!Consume('{')) return Fail("expected a JSON object");
    bool have_sites = false, have_samples = false;
    SkipWs();
    if (!Consume('}')) {
      while (true) {
        SkipWs();
        std::string key;
        if (!ParseString(&key)) return Fail("expected a member name");
        SkipWs();
        if (!Consume(':')) return Fail("expected ':'");
        SkipWs();
        if (key == "num_sites") {
          double v = 0;
          if (!ParseNumber(&v)) return Fail("num_sites is not a number");
          if (v < 0 || v > 4294967295.0 || v != std::floor(v))
            return "metadata num_sites is not an unsigned 32-bit integer";
          out->num_sites = (uint32_t)v;
          have_sites = true;
        } else if (key == "samples") {
          if (!Consume('[')) return Fail("samples is not an array");
          out->samples.clear();
          SkipWs();
          if (!Consume(']')) {
            while (true) {
              SkipWs();
              std::string id;
              if (!ParseString(&id)) return Fail("sample id is not a string");
              out->samples.push_back(std::move(id));
              SkipWs();
              if (Consume(',')) continue;
              if (Consume(']')) break;
              return Fail("expected ',' or ']'");
            }
          }
          have_samples = true;
        } else if (!SkipValue(0)) {
          return Fail("malformed value");
        }
        SkipWs();
        if (Consume(',')) continue;
        if (Consume('}')) break;
        return Fail("expected ',' or '}'");
      }
    }
    SkipWs();
    if (pos_ != s_.size()) return Fail("trailing characters");
    if (!have_samples) return "metadata has no \"samples\" array";
    if (!have_sites) return "metadata has no \"num_sites\"";
    return "";
  }

 private:
  std::string Fail(const char *what) const {
    std::ostringstream os;
    os << "Failed to parse metadata JSON: " << what << " at byte " << pos_;
    return os.str();
  }
  void SkipWs() {
    while (pos_ < s_.size() && (s_[pos_] == ' ' || s_[pos_] == '\t' ||
                                s_[pos_] == '\n' || s_[pos_] == '\r'))
      ++pos_;
  }
  bool Consume(char c) {
    if (pos_ < s_.size() && s_[pos_] == c) {
      ++pos_;
      return true;
    }
    return false;
  }
  static void AppendUtf8(uint32_t cp, std::string *out) {
    if (cp < 0x80) {
      out->push_back((char)cp);
    } else if (cp < 0x800) {
      out->push_back((char)(0xC0 | (cp >> 6)));
      out->push_back((char)(0x80 | (cp & 0x3F)));
    } else if (cp < 0x10000) {
      out->push_back((char)(0xE0 | (cp >> 12)));
      out->push_back((char)(0x80 | ((cp >> 6) & 0x3F)));
      out->push_back((char)(0x80 | (cp & 0x3F)));
    } else {
      out->push_back((char)(0xF0 | (cp >> 18)));
      out->push_back((char)(0x80 | ((cp >> 12) & 0x3F)));
      out->push_back((char)(0x80 | ((cp >> 6) & 0x3F)));
      out->push_back((char)(0x80 | (cp & 0x3F)));
    }
  }
  bool ParseHex4(uint32_t *out) {
    if (pos_ + 4 > s_.size()) return false;
    uint32_t v = 0;
    for (int k = 0; k < 4; ++k) {
      const char c = s_[pos_++];
      v <<= 4;
      if (c >= '0' && c <= '9') v |= (uint32_t)(c - '0');
      else if (c >= 'a' && c <= 'f') v |= (uint32_t)(c - 'a' + 10);
      else if (c >= 'A' && c <= 'F') v |= (uint32_t)(c - 'A' + 10);
      else return false;
    }
    *out = v;
    return true;
  }
  bool ParseString(std::string *out) {
    if (!Consume('"')) return false;
    out->clear();
    while (pos_ < s_.size()) {
      const unsigned char c = (unsigned char)s_[pos_++];
      if (c == '"') return true;
      if (c < 0x20) return false;
      if (c != '\\') {
        out->push_back((char)c);
        continue;
      }
      if (pos_ >= s_.size()) return false;
      const char e = s_[pos_++];
      switch (e) {
        case '"': out->push_back('"'); break;
        case '\\': out->push_back('\\'); break;
        case '/': out->push_back('/'); break;
        case 'b': out->push_back('\b'); break;
        case 'f': out->push_back('\f'); break;
        case 'n': out->push_back('\n'); break;
        case 'r': out->push_back('\r'); break;
        case 't': out->push_back('\t'); break;
        case 'u': {
          uint32_t cp = 0;
          if (!ParseHex4(&cp)) return false;
          if (cp >= 0xD800 && cp <= 0xDBFF) {  // surrogate pair
            uint32_t lo = 0;
            if (pos_ + 2 > s_.size() || s_[pos_] != '\\' || s_[pos_ + 1] != 'u')
              return false;
            pos_ += 2;
            if (!ParseHex4(&lo) || lo < 0xDC00 || lo > 0xDFFF) return false;
            cp = 0x10000 + ((cp - 0xD800) << 10) + (lo - 0xDC00);
          }
          AppendUtf8(cp, out);
          break;
        }
        default:
          return false;
      }
    }
    return false;
  }
  bool ParseNumber(double *out) {
    const size_t start = pos_;
    if (pos_ < s_.size() && s_[pos_] == '-') ++pos_;
    bool digits = false;
    while (pos_ < s_.size() &&
           ((s_[pos_] >= '0' && s_[pos_] <= '9') || s_[pos_] == '.' ||
            s_[pos_] == 'e' || s_[pos_] == 'E' || s_[pos_] == '+' ||
            s_[pos_] == '-')) {
      digits = digits || (s_[pos_] >= '0' && s_[pos_] <= '9');
      ++pos_;
    }
    if (!digits) return false;
    const std::string tok = s_.substr(start, pos_ - start);
    char *end = nullptr;
    *out = strtod(tok.c_str(), &end);
    return end != nullptr && *end == '\0';
  }
  bool ConsumeWord(const char *w) {
    size_t n = 0;
    while (w[n]) ++n;
    if (s_.compare(pos_, n, w) != 0) return false;
    pos_ += n;
    return true;
  }
  bool SkipValue(int depth) {
    if (depth > 64 || pos_ >= s_.size()) return false;
    const char c = s_[pos_];
    if (c == '"') {
      std::string tmp;
      return ParseString(&tmp);
    }
    if (c == '{' || c == '[') {
      const char close = c == '{' ? '}' : ']';
      ++pos_;
      SkipWs();
      if (Consume(close)) return true;
      while (true) {
        SkipWs();
        if (c == '{') {
          std::string key;
          if (!ParseString(&key)) return false;
          SkipWs();
          if (!Consume(':')) return false;
          SkipWs();
        }
        if (!SkipValue(depth + 1)) return false;
        SkipWs();
        if (Consume(',')) continue;
        return Consume(close);
      }
    }
    if (ConsumeWord("true") || ConsumeWord("false") || ConsumeWord("null"))
      return true;
    double v;
    return ParseNumber(&v);
  }

  const std::string &s_;
  size_t pos_ = 0;
};

}  // namespace

std::string ParseMetadata(const std::string &json_text, Metadata *out) {
  return Parser(json_text).Run(out);
}

std::string ReadMetadataFile(const std::string &path, Metadata *out) {
  std::ifstream in(path, std::ios::binary);
  if (!in) return "Failed to read metadata: cannot open " + path;  // cuking.cu:478-481
  std::ostringstream buf;
  buf << in.rdbuf();
  return ParseMetadata(buf.str(), out);
}

}  // namespace cuking_host
